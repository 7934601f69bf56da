#!/bin/bash
# A/B: does it matter in which phase of their tiles the two workgroups of a CU run relative to each other?  (second workgroup
# of every CU started 3 / 5 / 8 thousand cycles late; a tile period is ~10.8 thousand)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04g; rm -rf $O; mkdir -p $O
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
  for v in base stag3 stag5 stag8; do
    L=""; [ $v != base ] && L=$PWD/tools/exp/libexp_$v.so
    MFM_LIB=$L timeout 300 python bench.py $B --steps 200 --warmup 10 > $O/c64_${v}_$rep.json 2> $O/c64_${v}_$rep.err; summ $O/c64_${v}_$rep.json
  done
done
