#!/bin/bash
# SQ counters of the float path's kernel (two passes, --pmc with --kernel-trace only)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmcf
timeout 240 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmcf/p1 -o p -- python3 tools/bench_f32.py --iters 3 > gpurun_out/pmcf/p1.log 2>&1
timeout 240 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d gpurun_out/pmcf/p2 -o p -- python3 tools/bench_f32.py --iters 3 > gpurun_out/pmcf/p2.log 2>&1
python3 - <<'PY'
import csv, collections, os
for p in ("p1", "p2"):
    f = os.path.join("gpurun_out/pmcf", p, "p_counter_collection.csv")
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "f32_channel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("## pass", p)
    for k, v in sorted(acc.items()):
        print(f"{k:28s} launches={len(v):2d} mean={sum(v) / len(v):.6g}")
PY
