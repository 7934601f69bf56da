#!/bin/bash
# round 4, first GPU call: the whole GPU suite with the coalescing engine, then the block-size series
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04a; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -15 $O/pytest.log
timeout 600 python - > $O/series.json 2> $O/series.err <<'PY'
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
    d = json.loads(open("gpurun_out/r04a/series.json").read().strip().splitlines()[-1])
    for r in d["series"]:
        for m in ("coalesced", "per_block"):
            print(r["block_samples"], r["blocks"], m, {k: (round(v, 3) if isinstance(v, float) else v) for k, v in r[m].items()})
except Exception as e:
    print("ERR", e)
PY
