#!/bin/bash
# occupancy: the headline launch with one workgroup per CU (two waves per SIMD) against two (four): how far from saturation
# four waves are
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04o; rm -rf $O; mkdir -p $O
summ() { python3 - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print(sys.argv[1].split('/')[-1], "%.4g"%d["value"], "ms/step %.4f"%d["ms_per_step"], "kernel %.4f (min %.4f med %.4f)"%(r["kernel_ms"], r["kernel_ms_min"], r["kernel_ms_median"]), "grid", d["geometry"]["grid"], "verified", d.get("verified"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
}
B="--no-cpu-baseline --no-fp32 --no-chain --no-series"
for rep in 1 2; do
  for v in base onewg; do
    L=""; [ $v != base ] && L=$PWD/tools/exp/libexp_$v.so
    MFM_LIB=$L timeout 300 python bench.py $B --steps 100 --warmup 10 > $O/c64_${v}_$rep.json 2> $O/c64_${v}_$rep.err; summ $O/c64_${v}_$rep.json
    MFM_LIB=$L timeout 300 python bench.py $B --steps 20 --warmup 3 --config cfg3_1024ch --channels-per-gpu 1024 > $O/c1024_${v}_$rep.json 2> $O/c1024_${v}_$rep.err; summ $O/c1024_${v}_$rep.json
  done
done
