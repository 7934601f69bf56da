#!/bin/bash
# round 6, last call: the whole gpu suite on the tree as committed (with the driver's N > 1 launch form over the fake transport),
# smoke(), and the driver's bench command once more
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06g; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.txt
tail -3 $O/pytest_gpu.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r06g/bench_default.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["verified"], d["north_star_shape"]["kernel_ms"],
      {k: v.get("kernel_ms") for k, v in d.get("other_geometries", {}).items() if isinstance(v, dict)})
PY
