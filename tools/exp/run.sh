#!/bin/bash
# A/B of experiment libraries on one box: tools/exp/run.sh <masks...>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp
for rep in 1 2; do
for x in "$@"; do
  echo -n "X=$x: "
  MFM_LIB=$PWD/tools/exp/libexp_$x.so timeout 120 python bench.py --no-cpu-baseline --no-fp32 --steps 100 --warmup 10 --settle-seconds 0.5 ${BENCH_ARGS} 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(r['kernel_ms']*1000,1), round(r['kernel_ms_min']*1000,1), round(r['kernel_ms_median']*1000,1), round(d['ms_per_step']*1000,1))"
done; done
