#!/bin/bash
# round 4, fourth GPU call: whole GPU suite on the one-residual-step division, headline A/B against the previous library
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04d; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -6 $O/pytest.log
summ() { python3 - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print(sys.argv[1].split('/')[-1], r["kernel"], "%.4g"%d["value"], "ms/step %.4f"%d["ms_per_step"], "kernel %.4f (min %.4f med %.4f)"%(r["kernel_ms"], r["kernel_ms_min"], r["kernel_ms_median"]), "frac %.3f"%r["frac"], "verified", d.get("verified"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
}
B="--no-cpu-baseline --no-fp32 --no-chain --no-series"
for rep in 1 2 3; do
  for v in new old; do
    L=""; [ $v = old ] && L=$PWD/tools/exp/libprev.so
    MFM_LIB=$L timeout 300 python bench.py $B --steps 200 --warmup 10 > $O/ab_${v}_$rep.json 2> $O/ab_${v}_$rep.err; summ $O/ab_${v}_$rep.json
  done
done
for c in cfg2_64ch_grid; do timeout 300 python bench.py $B --steps 200 --warmup 10 --config $c > $O/grid.json 2> $O/grid.err; summ $O/grid.json; done
timeout 300 python bench.py $B --steps 40 --warmup 5 --config cfg3_1024ch --channels-per-gpu 1024 > $O/c1024.json 2> $O/c1024.err; summ $O/c1024.json
