#!/bin/bash
# round 3, first GPU call: whole GPU suite, then the headline bench (driver's flags, defaults), A/B of the per-wave exact-rotator
# path, the exact-grid plans and the many-channel shapes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03a; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -5 $O/pytest.log
B="--no-fp32 --no-chain"
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driverflags.json 2> $O/bench_driverflags.err; echo "rc=$?"
summ() { python3 - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print(sys.argv[1].split('/')[-1], r["kernel"], "%.4g"%d["value"], "ms/step %.4f"%d["ms_per_step"], "kernel %.4f (min %.4f med %.4f p95 %.4f)"%(r["kernel_ms"], r["kernel_ms_min"], r["kernel_ms_median"], r["kernel_ms_p95"]), "frac %.3f"%r["frac"], "verified", d.get("verified"), d.get("rotators"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
}
summ $O/bench_driverflags.json
for rep in 1 2; do
  for v in lib nowx; do
    L=""; [ $v = nowx ] && L=$PWD/tools/exp/libexp_nowx.so
    MFM_LIB=$L timeout 300 python bench.py --no-cpu-baseline $B --steps 100 --warmup 10 > $O/ab_${v}_$rep.json 2> $O/ab_${v}_$rep.err; summ $O/ab_${v}_$rep.json
  done
done
timeout 300 python bench.py --config cfg2_64ch_grid --no-cpu-baseline $B --steps 100 --warmup 10 > $O/bench_grid64.json 2> $O/bench_grid64.err; summ $O/bench_grid64.json
for c in 1024; do
  timeout 600 python bench.py --config cfg3_1024ch --channels-per-gpu $c --steps 40 --warmup 5 --no-cpu-baseline $B > $O/bench_c$c.json 2> $O/bench_c$c.err; summ $O/bench_c$c.json
  timeout 600 python bench.py --config cfg3_1024ch_grid --channels-per-gpu $c --steps 40 --warmup 5 --no-cpu-baseline $B > $O/bench_grid$c.json 2> $O/bench_grid$c.err; summ $O/bench_grid$c.json
done
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; summ $O/bench_default.json
