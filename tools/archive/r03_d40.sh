#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03d40; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "multiples_of_8 or reference_shaped or odd_geometries or 8bit_blocks or cfg2_64" > $O/pytest.log 2>&1; tail -15 $O/pytest.log
B="--no-fp32 --no-chain --no-cpu-baseline"
summ() { python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
print(sys.argv[2], r['kernel'], 'ms/step %.4f kernel %.4f (min %.4f med %.4f) frac %.3f verified %s' % (d['ms_per_step'], r['kernel_ms'], r['kernel_ms_min'], r['kernel_ms_median'], r['frac'], d.get('verified')))
" $1 "$2" 2>/dev/null || echo "$2 ERR $(tail -3 ${1%.json}.err)"; }
for rep in 1 2; do
for k in auto mfma1; do
  timeout 300 python bench.py $B --config multifm_1ch --channels-per-gpu 64 --kernel $k --steps 60 --warmup 10 > $O/d40_${k}_$rep.json 2> $O/d40_${k}_$rep.err; summ $O/d40_${k}_$rep.json "d40 $k"
done; done
