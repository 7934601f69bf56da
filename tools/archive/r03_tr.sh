#!/bin/bash
# first-generation kernel: PCM transposed through LDS (one 8-byte store per lane) against four 2-byte stores, same box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03tr; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | grep -E "passed|failed" | tail -2
B="--no-fp32 --no-chain --no-cpu-baseline --steps 40 --warmup 20"
lib() { [ "$1" = lib ] && echo "" || echo $PWD/tools/exp/libexp_$1.so; }
summ() { python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
print(sys.argv[2], 'ms/step %.4f kernel %.4f (min %.4f med %.4f) verified %s lds %s grid %s' % (d['ms_per_step'], r['kernel_ms'], r['kernel_ms_min'], r['kernel_ms_median'], d.get('verified'), d['geometry']['lds_bytes'], d['geometry']['grid']))
" $1 "$2" 2>/dev/null || echo "$2 ERR $(tail -2 ${1%.json}.err)"; }
for rep in 1 2 3; do for v in lib notr; do
  MFM_LIB=$(lib $v) timeout 300 python bench.py $B --config pocsag_rtlsdr --channels-per-gpu 64 > $O/d25_${v}_$rep.json 2> $O/d25_${v}_$rep.err; summ $O/d25_${v}_$rep.json "D25 64ch $v"
  MFM_LIB=$(lib $v) timeout 300 python bench.py $B --config cfg5_airspy --channels-per-gpu 256 > $O/c5_${v}_$rep.json 2> $O/c5_${v}_$rep.err; summ $O/c5_${v}_$rep.json "cfg5 256ch $v"
  MFM_LIB=$(lib $v) timeout 300 python bench.py $B --kernel mfma1 > $O/m1_${v}_$rep.json 2> $O/m1_${v}_$rep.err; summ $O/m1_${v}_$rep.json "cfg2 gen-1 $v"
done; done
