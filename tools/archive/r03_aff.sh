#!/bin/bash
# slice affinity (an XCD keeps its slices' rotator tables in L2) against the chunk-major item map, 512 / 1024 channels, same box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03aff; mkdir -p $O
B="--no-fp32 --no-chain --no-cpu-baseline --steps 40 --warmup 5"
lib() { [ "$1" = lib ] && echo "" || echo $PWD/tools/exp/libexp_$1.so; }
echo -n "parity lib: "; timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "many_channels or 1024 or bench_shape" 2>&1 | tail -1
summ() { python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
print(sys.argv[2], 'ms/step %.4f kernel %.4f (min %.4f med %.4f) verified %s traffic %s' % (d['ms_per_step'], r['kernel_ms'], r['kernel_ms_min'], r['kernel_ms_median'], d.get('verified'), r.get('traffic')))
" $1 "$2" 2>/dev/null || echo "$2 ERR $(tail -2 ${1%.json}.err)"; }
for rep in 1 2 3; do
for v in lib noaff; do
  for c in 1024 512; do
  MFM_LIB=$(lib $v) timeout 300 python bench.py $B --config cfg3_1024ch --channels-per-gpu $c > $O/${v}_${c}_$rep.json 2> $O/${v}_${c}_$rep.err; summ $O/${v}_${c}_$rep.json "$c ch $v"
  done
done; done
