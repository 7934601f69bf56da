#!/bin/bash
# first-generation kernels: what the arctangent-table reads and the wait behind them cost (timing build without them, wrong PCM)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04l; rm -rf $O; mkdir -p $O
summ() { python3 - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print(sys.argv[1].split('/')[-1], "kernel %.4f (min %.4f med %.4f)"%(r["kernel_ms"], r["kernel_ms_min"], r["kernel_ms_median"]), "verified", d.get("verified"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
}
B="--no-cpu-baseline --no-fp32 --no-chain --no-series"
for rep in 1 2; do
  for v in base nolut; do
    L=""; [ $v != base ] && L=$PWD/tools/exp/libexp_$v.so
    MFM_LIB=$L timeout 300 python bench.py $B --config cfg5_airspy --channels-per-gpu 256 --steps 40 --warmup 5 > $O/cfg5_${v}_$rep.json 2> $O/cfg5_${v}_$rep.err; summ $O/cfg5_${v}_$rep.json
    MFM_LIB=$L timeout 300 python bench.py $B --config pocsag_rtlsdr_256taps --channels-per-gpu 64 --steps 40 --warmup 5 > $O/d25_${v}_$rep.json 2> $O/d25_${v}_$rep.err; summ $O/d25_${v}_$rep.json
    MFM_LIB=$L timeout 300 python bench.py $B --kernel mfma1 --steps 100 --warmup 5 > $O/m1_${v}_$rep.json 2> $O/m1_${v}_$rep.err; summ $O/m1_${v}_$rep.json
  done
done
