#!/bin/bash
# round 4, ninth GPU call: decimation 25 on the second-generation kernel (padded rows): parity, then time against the first generation
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04i; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_coalesce.py tests/test_host.py tests/test_pocsag.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -12 $O/pytest.log
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
  timeout 300 python bench.py $B --steps 60 --warmup 5 --config pocsag_rtlsdr --channels-per-gpu 64 > $O/d25_v3_$rep.json 2> $O/d25_v3_$rep.err; summ $O/d25_v3_$rep.json
  timeout 300 python bench.py $B --steps 60 --warmup 5 --config pocsag_rtlsdr --channels-per-gpu 64 --kernel mfma1 > $O/d25_v1_$rep.json 2> $O/d25_v1_$rep.err; summ $O/d25_v1_$rep.json
done
timeout 300 python tools/bench_ingest8.py --help > /dev/null 2>&1
