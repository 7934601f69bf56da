#!/bin/bash
# A/B: first-generation kernels with both channels' discriminators together (one wait for the table) against one after the other
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04d; rm -rf $O; mkdir -p $O
summ() { python3 - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print(sys.argv[1].split('/')[-1], "kernel %.4f (min %.4f med %.4f)"%(r["kernel_ms"], r["kernel_ms_min"], r["kernel_ms_median"]), "verified", d.get("verified"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
}
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_group.py -m gpu -q -x 2>&1 | grep "passed\|failed\|Error" | tail -2
timeout 300 python tools/fuzz_engine.py --long --seconds 150 --seed 51 2>&1 | tail -1
B="--no-cpu-baseline --no-fp32 --no-chain --no-series"
for rep in 1 2 3; do
  for v in base disc2; do
    L=""; [ $v != base ] && L=$PWD/tools/exp/libexp_$v.so
    MFM_LIB=$L timeout 300 python bench.py $B --config cfg5_airspy --channels-per-gpu 256 --steps 40 --warmup 5 > $O/cfg5_${v}_$rep.json 2> $O/cfg5_${v}_$rep.err; summ $O/cfg5_${v}_$rep.json
    MFM_LIB=$L timeout 300 python bench.py $B --config pocsag_rtlsdr_256taps --channels-per-gpu 64 --steps 40 --warmup 5 > $O/d25_${v}_$rep.json 2> $O/d25_${v}_$rep.err; summ $O/d25_${v}_$rep.json
    MFM_LIB=$L timeout 300 python bench.py $B --kernel mfma1 --steps 100 --warmup 5 > $O/m1_${v}_$rep.json 2> $O/m1_${v}_$rep.err; summ $O/m1_${v}_$rep.json
    MFM_LIB=$L timeout 300 python bench.py $B --config cfg2_64ch_512taps --steps 40 --warmup 5 > $O/t512_${v}_$rep.json 2> $O/t512_${v}_$rep.err; summ $O/t512_${v}_$rep.json
  done
done
