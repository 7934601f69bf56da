#!/bin/bash
# A/B of library variants (tools/exp/variant.sh) on one box: tools/exp/run.sh <names...>; BENCH_ARGS = extra bench.py flags
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp
for rep in $(seq 1 ${AB_REPS:-3}); do
for x in "$@"; do
  echo -n "X=$x ${BENCH_ARGS}: "
  MFM_LIB=$PWD/tools/exp/libexp_$x.so timeout 200 python bench.py --no-cpu-baseline --no-fp32 --no-chain --no-series --steps 200 --warmup 20 --settle-seconds 0.5 ${BENCH_ARGS} 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; c=r.get('clocks',{}); print(round(r['kernel_ms']*1000,1), round(r['kernel_ms_min']*1000,1), round(r['kernel_ms_median']*1000,1), round(d['ms_per_step']*1000,1), 'verified', d.get('verified'), 'cycles', c.get('shader_ticks_median'), round(c.get('sclk_mhz_effective') or 0))"
done; done
