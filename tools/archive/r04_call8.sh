#!/bin/bash
# round 4, eighth GPU call: whole GPU suite on the shipped build (nt PCM stores, 4-byte rotator entries), the raster plans beside the
# general ones, other geometries with the first-generation kernel's stores hinted non-temporal (A/B)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04h; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
summ() { python3 - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print(sys.argv[1].split('/')[-1], "%.4g"%d["value"], "ms/step %.4f"%d["ms_per_step"], "kernel %.4f (min %.4f med %.4f)"%(r["kernel_ms"], r["kernel_ms_min"], r["kernel_ms_median"]), "frac %.3f"%r["frac"], "verified", d.get("verified"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
}
B="--no-cpu-baseline --no-fp32 --no-chain --no-series"
for rep in 1 2; do
  for c in cfg3_1024ch cfg3_1024ch_grid; do
    timeout 300 python bench.py $B --steps 40 --warmup 5 --config $c --channels-per-gpu 1024 > $O/${c}_$rep.json 2> $O/${c}_$rep.err; summ $O/${c}_$rep.json
  done
done
for rep in 1 2; do
  for v in base m1nt; do
    L=""; [ $v != base ] && L=$PWD/tools/exp/libexp_$v.so
    MFM_LIB=$L timeout 300 python bench.py $B --steps 60 --warmup 5 --config pocsag_rtlsdr --channels-per-gpu 64 > $O/d25_${v}_$rep.json 2> $O/d25_${v}_$rep.err; summ $O/d25_${v}_$rep.json
    MFM_LIB=$L timeout 300 python bench.py $B --steps 60 --warmup 5 --config cfg5_airspy --channels-per-gpu 256 > $O/cfg5_${v}_$rep.json 2> $O/cfg5_${v}_$rep.err; summ $O/cfg5_${v}_$rep.json
  done
done
