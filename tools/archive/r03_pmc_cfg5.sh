#!/bin/bash
# SQ counters of the configs[4] share (256 ch, D = 400, 512 taps): resident taps against streamed taps, one box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03pmc5; rm -rf $O; mkdir -p $O
for k in auto mfma1s; do
P="python3 bench.py --config cfg5_airspy --channels-per-gpu 256 --kernel $k --steps 8 --warmup 3 --settle-seconds 0.3 --no-cpu-baseline --no-fp32 --no-chain"
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/${k}_p1 -o p -- $P > $O/${k}_p1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_MFMA --kernel-trace --output-format csv -d $O/${k}_p2 -o p -- $P > $O/${k}_p2.log 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT --kernel-trace --output-format csv -d $O/${k}_p3 -o p -- $P > $O/${k}_p3.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${k}_f -o p -- $P > $O/${k}_f.log 2>&1
done
python3 - <<'PY'
import csv, collections, glob, os
out = []
for k in ("auto", "mfma1s"):
    acc = collections.defaultdict(list)
    for p in ("p1", "p2", "p3", "f"):
        for f in glob.glob(f"gpurun_out/r03pmc5/{k}_{p}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "mfm_channel_kernel" in r["Kernel_Name"]:
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    out.append("== %s ==" % ("taps resident" if k == "auto" else "taps streamed (MFM_F_STREAM_TAPS)"))
    m = {}
    for c, v in sorted(acc.items()):
        v = v[len(v) // 2:]
        m[c] = sum(v) / len(v)
        out.append(f"{c:30s} launches={len(v):2d} mean={m[c]:.6g}")
    cyc = m["GRBM_GUI_ACTIVE"] / 8.0
    other = m["SQ_INSTS_VALU"] - m["SQ_INSTS_MFMA"]
    out.append(f"  launch length {cyc:.4g} cycles; matrix pipe busy {100 * m['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * cyc):.1f} % of SIMD cycles; "
               f"other VALU {other:.4g} instructions = {100 * 3 * other / (1024 * cyc):.1f} % at 3 cycles each; "
               f"SALU {m['SQ_INSTS_SALU']:.3g}, LDS {m['SQ_INSTS_LDS']:.3g}, VMEM reads {m['SQ_INSTS_VMEM_RD']:.3g} instructions")
    out.append(f"  waves: issuing {100 * m['SQ_ACTIVE_INST_ANY'] / m['SQ_WAVE_CYCLES']:.0f} %, waiting for an issue slot "
               f"{100 * m['SQ_WAIT_INST_ANY'] / m['SQ_WAVE_CYCLES']:.0f} %, waiting for data {100 * m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES']:.0f} % of their cycles; "
               f"HBM reads {2 * m['FETCH_SIZE'] * 1024 / 1e6:.0f} MB per launch (2 x FETCH_SIZE)")
open("gpurun_out/r03pmc5/summary.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
