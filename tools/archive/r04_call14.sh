#!/bin/bash
# A/B: do the resident sixteen-/eight-k-step instances pay for carrying the per-sample (split-row) staging path?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04s; rm -rf $O; mkdir -p $O
summ() { python3 - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print(sys.argv[1].split('/')[-1], "%.4g"%d["value"], "ms/step %.4f"%d["ms_per_step"], "kernel %.4f (min %.4f med %.4f)"%(r["kernel_ms"], r["kernel_ms_min"], r["kernel_ms_median"]), "verified", d.get("verified"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
}
B="--no-cpu-baseline --no-fp32 --no-chain --no-series"
for rep in 1 2 3; do
  for v in base nosplit; do
    L=""; [ $v != base ] && L=$PWD/tools/exp/libexp_$v.so
    MFM_LIB=$L timeout 300 python bench.py $B --config cfg5_airspy --channels-per-gpu 256 --steps 40 --warmup 5 > $O/cfg5_${v}_$rep.json 2> $O/cfg5_${v}_$rep.err; summ $O/cfg5_${v}_$rep.json
    MFM_LIB=$L timeout 300 python bench.py $B --config cfg2_64ch_512taps --steps 40 --warmup 5 > $O/t512_${v}_$rep.json 2> $O/t512_${v}_$rep.err; summ $O/t512_${v}_$rep.json
    MFM_LIB=$L timeout 300 python bench.py $B --config cfg2_64ch_256taps --steps 40 --warmup 5 > $O/t256_${v}_$rep.json 2> $O/t256_${v}_$rep.err; summ $O/t256_${v}_$rep.json
  done
done
