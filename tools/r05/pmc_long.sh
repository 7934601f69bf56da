#!/bin/bash
# Round 5: SQ counters of a long-filter shape, second-generation long-filter kernel (auto) against the first generation (mfma1),
# one box.  tools/r05/pmc_long.sh <plan> <channels> [more bench.py flags]   e.g. cfg5_airspy 256
PLAN=${1:-cfg5_airspy}; NCH=${2:-256}; shift 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05/pmc_${PLAN}_${NCH}; rm -rf $O; mkdir -p $O
for k in auto mfma1 v3l1; do
P="python3 bench.py --config $PLAN --channels-per-gpu $NCH --kernel $k --steps 8 --warmup 3 --settle-seconds 0.3 --no-cpu-baseline --no-fp32 --no-chain --no-series $*"
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/${k}_p1 -o p -- $P > $O/${k}_p1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_MFMA --kernel-trace --output-format csv -d $O/${k}_p2 -o p -- $P > $O/${k}_p2.log 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT --kernel-trace --output-format csv -d $O/${k}_p3 -o p -- $P > $O/${k}_p3.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${k}_t -o p -- $P > $O/${k}_t.log 2>&1
done
python3 - "$O" "$PLAN" "$NCH" <<'PY'
import csv, collections, glob, json, sys
O, plan, nch = sys.argv[1], sys.argv[2], int(sys.argv[3])
out = [f"# tools/r05/pmc_long.sh {plan} {nch}: rocprofv3 --pmc passes (SQ counters in three passes) and a --kernel-trace --stats pass of",
       f"# bench.py --config {plan} --channels-per-gpu {nch} --kernel <k> --steps 8 --warmup 3 --settle-seconds 0.3, block 2^26;",
       "# mean per launch over the second half of the profiled launches."]
for k, what in (("auto", "second generation, long-filter kernel (mfm_kernel_v3l.hip)"), ("v3l1", "the same, one row block per wave forced"),
                ("mfma1", "first generation, taps resident (MFM_F_FORCE_MFMA_V1)")):
    acc = collections.defaultdict(list)
    name = None
    for p in ("p1", "p2", "p3"):
        for f in glob.glob(f"{O}/{k}_{p}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "mfm_channel_kernel" in r["Kernel_Name"]:
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
                    name = r["Kernel_Name"].split("(")[0]
    dur = None
    for f in glob.glob(f"{O}/{k}_t/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "mfm_channel_kernel" in r["Name"]:
                dur = float(r["AverageNs"]) / 1e3
    bench = None
    try:
        bench = json.loads([ln for ln in open(f"{O}/{k}_t.log") if ln.startswith("{")][-1])
    except Exception:
        pass
    out.append(f"== {what} ==  {name}")
    m = {}
    for c, v in sorted(acc.items()):
        v = v[len(v) // 2:]
        m[c] = sum(v) / len(v)
        out.append(f"{c:30s} launches={len(v):3d} mean={m[c]:.6g}")
    if not m:
        out.append("  (no counters)")
        continue
    cyc = m["GRBM_GUI_ACTIVE"] / 8.0
    other = m["SQ_INSTS_VALU"] - m["SQ_INSTS_MFMA"]
    outs = nch * ((1 << 26) // (bench["config"].get("decimation", 0) or 1)) if bench and "config" in bench else 0
    out.append(f"  launch length {cyc:.4g} cycles; matrix pipe busy {100 * m['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * cyc):.1f} % of SIMD cycles; "
               f"other VALU {other:.4g} wave instructions = {100 * 3 * other / (1024 * cyc):.1f} % at 3 cycles each; "
               f"SALU {m['SQ_INSTS_SALU']:.3g}, LDS {m['SQ_INSTS_LDS']:.3g}, VMEM reads {m['SQ_INSTS_VMEM_RD']:.3g}, MFMA {m['SQ_INSTS_MFMA']:.4g} instructions")
    out.append(f"  waves: issuing {100 * m['SQ_ACTIVE_INST_ANY'] / m['SQ_WAVE_CYCLES']:.0f} %, waiting for an issue slot "
               f"{100 * m['SQ_WAIT_INST_ANY'] / m['SQ_WAVE_CYCLES']:.0f} %, waiting for data {100 * m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES']:.0f} % of their cycles; "
               f"LDS wait {100 * m['SQ_WAIT_INST_LDS'] / m['SQ_WAVE_CYCLES']:.1f} %; LDS bank conflict cycles {m['SQ_LDS_BANK_CONFLICT']:.3g} of {m['SQ_LDS_IDX_ACTIVE']:.3g} active")
    if dur:
        out.append(f"  rocprofv3 --kernel-trace --stats: average {dur:.1f} us per launch")
    if bench:
        out.append(f"  bench line: kernel_ms {bench['roofline']['kernel_ms']:.4f}, ms_per_step {bench['ms_per_step']:.4f}, verified {bench.get('verified')}")
        per = 64.0 * other / (nch * ((1 << 26) // bench['config']['decimation'])) if 'decimation' in bench.get('config', {}) else None
        if per:
            out.append(f"  non-matrix VALU lane-instructions per (channel, output): {per:.1f}")
open(f"{O}/summary.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
