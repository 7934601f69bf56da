#!/bin/bash
# resident long-filter instances against the streamed form, same box: parity first, then alternating bench runs
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03res; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "long_filters or airspy or single_iteration or odd_geometries or fuzz or not_multiples or resident" 2>&1 | tail -3
B="--no-fp32 --no-chain --no-cpu-baseline --steps 40 --warmup 20"
summ() { python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
print(sys.argv[2], 'ms/step %.4f kernel %s %.4f (min %.4f med %.4f) verified %s mfma frac %.3f' % (d['ms_per_step'], r['kernel'], r['kernel_ms'], r['kernel_ms_min'], r['kernel_ms_median'], d.get('verified'), d['compute_roofline']['frac']))
" $1 "$2" 2>/dev/null || echo "$2 ERR $(tail -2 ${1%.json}.err)"; }
for rep in 1 2; do
for k in auto mfma1s; do
  timeout 300 python bench.py $B --kernel $k --config cfg5_airspy --channels-per-gpu 256 > $O/a_${k}_$rep.json 2> $O/a_${k}_$rep.err; summ $O/a_${k}_$rep.json "cfg5 256ch $k"
  timeout 300 python bench.py $B --kernel $k --config cfg2_64ch_512taps > $O/b_${k}_$rep.json 2> $O/b_${k}_$rep.err; summ $O/b_${k}_$rep.json "512taps D96 $k"
  timeout 300 python bench.py $B --kernel $k --config cfg2_64ch_256taps > $O/c_${k}_$rep.json 2> $O/c_${k}_$rep.err; summ $O/c_${k}_$rep.json "256taps D96 $k"
done; done
