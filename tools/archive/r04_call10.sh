#!/bin/bash
# round 4, tenth GPU call: the two channels of a lane's epilogue interleaved by the compiler (no scheduling barrier between them)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04m; rm -rf $O; mkdir -p $O
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
for rep in 1 2 3; do
  for v in base sp4160 sp4288; do
    L=""; [ $v != base ] && L=$PWD/tools/exp/libexp_$v.so
    MFM_LIB=$L timeout 300 python bench.py $B --steps 200 --warmup 10 > $O/c64_${v}_$rep.json 2> $O/c64_${v}_$rep.err; summ $O/c64_${v}_$rep.json
  done
done
for rep in; do
  for v in base sp4160 sp4288; do
    L=""; [ $v != base ] && L=$PWD/tools/exp/libexp_$v.so
    MFM_LIB=$L timeout 300 python bench.py $B --steps 60 --warmup 5 --config pocsag_rtlsdr --channels-per-gpu 64 > $O/d25_${v}_$rep.json 2> $O/d25_${v}_$rep.err; summ $O/d25_${v}_$rep.json
  done
done
for v in sp4160 sp4288; do
  MFM_LIB=$PWD/tools/exp/libexp_$v.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "cfg2_64 or reference_shaped or golden or many_channels" 2>&1 | tail -2
done
timeout 300 python3 - <<'PY'
import json, sys, os
sys.path.insert(0, os.getcwd())
import bench
from __graft_entry__ import load_package
pkg = load_package()
fs, decim, taps, offs, gains = pkg.synth.plan("cfg2_64ch", nr_channels=64)
e = bench.end_to_end(pkg, fs, decim, taps, offs, gains)
for k, v in e.items():
    if isinstance(v, dict):
        print(k, {a: (round(b, 2) if isinstance(b, float) else b) for a, b in v.items()})
PY
