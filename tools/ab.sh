#!/bin/bash
# A/B timing of bench.py's kernel under different env settings on the same box: tools/ab.sh "VAR=a" "VAR=b" ...
for rep in $(seq 1 ${AB_REPS:-2}); do
  for cfg in "$@"; do
    echo -n "$cfg: "; env $cfg timeout 300 python bench.py --no-cpu-baseline --no-fp32 --no-chain --steps ${AB_STEPS:-40} 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'], d['ms_per_step'])"
  done
done
