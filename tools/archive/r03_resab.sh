#!/bin/bash
# A/B of library variants on the long-filter shapes, same box: tools/archive/r03_resab.sh <variant...>  ("lib" = the shipped library)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03resab; mkdir -p $O
B="--no-fp32 --no-chain --no-cpu-baseline --steps 40 --warmup 20"
lib() { [ "$1" = lib ] && echo "" || echo $PWD/tools/exp/libexp_$1.so; }
summ() { python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
print(sys.argv[2], 'ms/step %.4f kernel %.4f (min %.4f med %.4f) verified %s' % (d['ms_per_step'], r['kernel_ms'], r['kernel_ms_min'], r['kernel_ms_median'], d.get('verified')))
" $1 "$2" 2>/dev/null || echo "$2 ERR $(tail -2 ${1%.json}.err)"; }
for rep in 1 2 3; do
for v in "$@"; do
  MFM_LIB=$(lib $v) timeout 300 python bench.py $B --config cfg5_airspy --channels-per-gpu 256 > $O/a_${v}_$rep.json 2> $O/a_${v}_$rep.err; summ $O/a_${v}_$rep.json "cfg5 256ch $v"
  MFM_LIB=$(lib $v) timeout 300 python bench.py $B --config cfg2_64ch_512taps > $O/b_${v}_$rep.json 2> $O/b_${v}_$rep.err; summ $O/b_${v}_$rep.json "512taps D96 $v"
done; done
