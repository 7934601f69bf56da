#!/bin/bash
# Round 6, second call: the driver's command (and the default line) once more on the same library, now that the first call's
# counters are in profiles/r06_issue_model.json - the line then carries roofline.issue_model / ceiling_frac itself.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06f; rm -rf $O; mkdir -p $O
T0=$SECONDS; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driverflags.json 2> $O/bench_driverflags.err; echo "driver-flags run: $((SECONDS - T0)) s wall" > $O/driver_run_s.txt
timeout 400 python bench.py --no-fp32 --no-chain --no-series > $O/bench_default.json 2> $O/bench_default.err
python3 - <<'PY'
import json
l = json.loads(open("gpurun_out/r06f/bench_driverflags.json").read().strip().splitlines()[-1])
print(l["value"], l["ms_per_step"], json.dumps(l["roofline"].get("issue_model"))[:600])
print(json.dumps(l.get("north_star_shape", {}).get("issue_model"))[:600])
PY
