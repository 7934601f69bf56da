#!/bin/bash
# round 4, second GPU call: GPU suite with independent launches / overlap / seek, headline one stream against two, block series
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04b; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -12 $O/pytest.log
summ() { python3 - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print(sys.argv[1].split('/')[-1], r["kernel"], "%.4g"%d["value"], "ms/step %.4f"%d["ms_per_step"], "kernel %.4f (min %.4f med %.4f p95 %.4f)"%(r["kernel_ms"], r["kernel_ms_min"], r["kernel_ms_median"], r["kernel_ms_p95"]), "frac %.3f"%r["frac"], "verified", d.get("verified"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
}
B="--no-cpu-baseline --no-fp32 --no-chain"
for rep in 1 2 3; do
  timeout 300 python bench.py $B --steps 200 --warmup 10 > $O/one_$rep.json 2> $O/one_$rep.err; summ $O/one_$rep.json
  timeout 300 python bench.py $B --steps 200 --warmup 10 --overlap > $O/two_$rep.json 2> $O/two_$rep.err; summ $O/two_$rep.json
done
timeout 300 python bench.py $B --steps 20 --warmup 5 --overlap > $O/two_drv.json 2> $O/two_drv.err; summ $O/two_drv.json
timeout 300 python bench.py $B --steps 40 --warmup 5 --overlap --config cfg3_1024ch --channels-per-gpu 1024 > $O/two_1024.json 2> $O/two_1024.err; summ $O/two_1024.json
timeout 300 python bench.py $B --steps 40 --warmup 5 --config cfg3_1024ch --channels-per-gpu 1024 > $O/one_1024.json 2> $O/one_1024.err; summ $O/one_1024.json
timeout 900 python - > $O/series.json 2> $O/series.err <<'PY'
import json, sys, os
sys.path.insert(0, os.getcwd())
import torch, bench
from __graft_entry__ import load_package
pkg = load_package()
fs, decim, taps, offs, gains = pkg.synth.plan("cfg2_64ch", nr_channels=64)
print(json.dumps(bench.block_series(pkg, torch, fs, decim, taps, offs, gains)))
PY
echo "series rc=$?"; tail -3 $O/series.err
python3 - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r04b/series.json").read().strip().splitlines()[-1])
    for r in d["series"]:
        for m in r:
            if isinstance(r[m], dict):
                print(r["block_samples"], r["blocks"], m, {k: (round(v, 3) if isinstance(v, float) else v) for k, v in r[m].items()})
except Exception as e:
    print("ERR", e)
PY
