#!/bin/bash
# round 4, fifth GPU call: GPU suite (pinned pushes, host on the page-locked pool), default bench line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04e; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -6 $O/pytest.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"; tail -2 $O/bench_n1.err
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r04e/bench_n1.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("value %.4g ms/step %.4f kernel %.4f frac %.3f verified %s" % (d["value"], d["ms_per_step"], r["kernel_ms"], r["frac"], d["verified"]))
print(json.dumps(d["end_to_end"], indent=1))
print({k: v for k, v in d["other_geometries"].items()})
PY
