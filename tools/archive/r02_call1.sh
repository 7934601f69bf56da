#!/bin/bash
# round 2, first GPU call: issue-model microbenchmark, many-channel parity, bench lines at C = 64/128/256/1024
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02a; mkdir -p $O
timeout 300 ./tools/ubench_issue > $O/ubench_issue.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q -k "many_channels or 2048_channels or cfg2_64" > $O/pytest_many.txt 2>&1
tail -5 $O/pytest_many.txt
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-fp32 > $O/bench_driverflags.json 2> $O/bench_driverflags.err
tail -c 600 $O/bench_driverflags.json
timeout 300 python bench.py --no-cpu-baseline --no-fp32 > $O/bench_default.json 2> $O/bench_default.err
for c in 128 256 1024; do
  timeout 600 python bench.py --config cfg3_1024ch --channels-per-gpu $c --steps 30 --warmup 5 --no-cpu-baseline --no-fp32 > $O/bench_c$c.json 2> $O/bench_c$c.err
done
for f in $O/bench_*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d["roofline"]; print(d["value"], d["ms_per_step"], r["kernel_ms"], r.get("kernel_ms_min"), r.get("kernel_ms_median"), r.get("kernel_ms_p95"), r["frac"], d["compute_roofline"]["frac"])
except Exception as e:
    print("ERR", e)
PY
done
