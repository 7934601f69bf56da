#!/bin/bash
# cycles per launch (GRBM_GUI_ACTIVE) and SQ busy/wait counters for experiment libraries: tools/exp/pmc.sh <masks...>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp
for x in "$@"; do
  rm -rf gpurun_out/exp/pmc_$x
  MFM_LIB=$PWD/tools/exp/libexp_$x.so timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d gpurun_out/exp/pmc_$x -o p -- python3 bench.py --steps 20 --warmup 5 --settle-seconds 0.3 --no-cpu-baseline --no-fp32 > gpurun_out/exp/pmc_$x.log 2>&1
  python3 - "$x" <<'PY'
import csv,glob,sys,collections
x=sys.argv[1]
f=glob.glob(f"gpurun_out/exp/pmc_{x}/**/*counter_collection.csv",recursive=True)
acc=collections.defaultdict(list)
for fn in f:
    for r in csv.DictReader(open(fn)):
        if "channel_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("X=%s"%x, {k:"%.4g"%(sum(v[len(v)//2:])/max(1,len(v[len(v)//2:]))) for k,v in sorted(acc.items())})
PY
done
