cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc5; rm -rf $O; mkdir -p $O
P="python3 bench.py --config cfg5_airspy --channels-per-gpu 256 --steps 8 --warmup 3 --settle-seconds 0.3 --no-cpu-baseline --no-fp32 --no-chain"
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/p1 -o p -- $P > $O/p1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_MFMA --kernel-trace --output-format csv -d $O/p2 -o p -- $P > $O/p2.log 2>&1
python3 - <<'PY'
import csv, collections, os
for p in ("p1", "p2"):
    f = os.path.join("gpurun_out/pmc5", p, "p_counter_collection.csv")
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "mfm_channel_kernel" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(k[0], f"{k[1]:28s} launches={len(v):2d} mean={sum(v) / len(v):.6g}")
PY
