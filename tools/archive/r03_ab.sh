#!/bin/bash
# A/B of library variants on one box: tools/archive/r03_ab.sh <variant...>   ("lib" = the shipped library, else tools/exp/libexp_<v>.so)
# per variant: a quick parity run (cfg2 + mixed rotator classes), then alternating bench runs at 2^26 and 2^22-sample blocks
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03ab; mkdir -p $O
B="--no-fp32 --no-chain --no-cpu-baseline"
lib() { [ "$1" = lib ] && echo "" || echo $PWD/tools/exp/libexp_$1.so; }
for v in "$@"; do
  echo -n "parity $v: "; MFM_LIB=$(lib $v) timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "cfg2_64 or rotator_classes or ragged or long_stream or channel_counts" 2>&1 | tail -1
done
summ() { python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
print(sys.argv[2], 'ms/step %.4f kernel %.4f (min %.4f med %.4f p95 %.4f) verified %s' % (d['ms_per_step'], r['kernel_ms'], r['kernel_ms_min'], r['kernel_ms_median'], r['kernel_ms_p95'], d.get('verified')))
" $1 "$2" 2>/dev/null || echo "$2 ERR $(tail -2 ${1%.json}.err)"; }
for rep in 1 2 3; do
  for v in "$@"; do
    MFM_LIB=$(lib $v) timeout 300 python bench.py $B --steps 100 --warmup 10 ${AB_ARGS} > $O/${v}_$rep.json 2> $O/${v}_$rep.err; summ $O/${v}_$rep.json "$v 2^26"
  done
done
for rep in 1 2; do
  for v in "$@"; do
    MFM_LIB=$(lib $v) timeout 300 python bench.py $B --steps 200 --warmup 20 --block-log2 22 > $O/${v}_s$rep.json 2> $O/${v}_s$rep.err; summ $O/${v}_s$rep.json "$v 2^22"
  done
done
