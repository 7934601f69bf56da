#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04l; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; grep "f32 path" $O/pytest.log | head -20; tail -5 $O/pytest.log
timeout 300 python bench.py --no-cpu-baseline --no-fp32 --no-chain --no-series --steps 100 --warmup 10 --input rtlsdr_u8 > $O/in8.json 2> $O/in8.err; tail -2 $O/in8.err
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r04l/in8.json").read().strip().splitlines()[-1]); r = d["roofline"]
print("in8: value %.4g ms/step %.4f kernel %.4f frac %.3f verified %s" % (d["value"], d["ms_per_step"], r["kernel_ms"], r["frac"], d["verified"]))
PY
