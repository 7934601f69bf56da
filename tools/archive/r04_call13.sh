#!/bin/bash
# round 4: where the 30 % of SIMD cycles that issue nothing go - instruction fetch, LDS / VMEM queue depths and FIFO stalls
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04w; rm -rf $O; mkdir -p $O
B="--no-cpu-baseline --no-fp32 --no-chain --no-series"
P="python3 bench.py --steps 8 --warmup 3 --settle-seconds 0.3 $B"
pass() { n=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$n -o p -- $P > $O/$n.log 2>&1
  python3 - $O/$n $n <<'PY'
import csv, glob, sys, collections
d, v = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mfm_channel_kernel_v3" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, x in sorted(acc.items()):
    h = x[len(x)//2:]
    print(v, k, "%.5g" % (sum(h) / max(1, len(h))))
PY
}
pass w1 SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA
pass w2 SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_BRANCH SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT
pass w3 SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_ACTIVE_INST_VALU2 SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES
pass w4 SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_LEVEL_WAVES SQ_WAVES SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_LDS
