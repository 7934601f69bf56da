#!/bin/bash
# Round 5: where the 1024-channel shape's extra L2-miss traffic comes from (1.30-1.37 x algorithmic): FETCH_SIZE of the general
# plan against the raster plan (no rotator tables read), and the size of the tables.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05/traffic1024; rm -rf $O; mkdir -p $O
N="--no-cpu-baseline --no-fp32 --no-chain --no-series --steps 6 --warmup 2 --settle-seconds 0.3 --channels-per-gpu 1024"
for plan in cfg3_1024ch cfg3_1024ch_grid; do
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${plan}_fetch -o f -- python3 bench.py --config $plan $N > $O/${plan}_fetch.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${plan}_write -o w -- python3 bench.py --config $plan $N > $O/${plan}_write.log 2>&1
done
python3 - <<'PY'
import csv, glob, json, sys, os
sys.path.insert(0, os.getcwd())
O = "gpurun_out/r05/traffic1024"
for plan in ("cfg3_1024ch", "cfg3_1024ch_grid"):
    res = {}
    for name, d in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
        for f in glob.glob(f"{O}/{plan}_{d}/**/*counter_collection.csv", recursive=True):
            v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "channel_kernel" in r["Kernel_Name"] and r["Counter_Name"] == name]
            v = v[len(v) // 2:]
            res[name] = sum(v) / max(1, len(v))
    print(plan, {k: round(v) for k, v in res.items()}, "KB ->", round((2 * res.get("FETCH_SIZE", 0) + res.get("WRITE_SIZE", 0)) / 1024), "MB per launch (algorithmic 1700 MB: 268 in + 1432 PCM)")
from __graft_entry__ import load_package
pkg = load_package()
for plan in ("cfg3_1024ch", "cfg3_1024ch_grid"):
    fs, decim, taps, offs, gains = pkg.synth.plan(plan, nr_channels=1024)
    eng = pkg.Engine(fs, decim, 1 << 20, device=0, flags=pkg.binding.MFM_F_DEVICE_ONLY)
    for o, g in zip(offs, gains):
        eng.add_channel(int(o), taps, float(g))
    eng.commit()
    st = eng.stats()
    mus = [eng.get_channel_info(c) if hasattr(eng, "get_channel_info") else None for c in range(0)]
    print(plan, "rotator table entries", st["rot_table_entries"], "=", st["rot_table_entries"] * 4 / 1e6, "MB at 4 bytes; exact channels", st["rot_exact_channels"])
    eng.close()
PY
