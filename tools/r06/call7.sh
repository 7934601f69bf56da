#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python tools/r05/v3l_smoke.py quick > gpurun_out/r06/v3l_smoke.txt 2>&1; tail -1 gpurun_out/r06/v3l_smoke.txt
timeout 500 python tools/r06/repro.py > gpurun_out/r06/repro.txt 2>&1; grep -c " OK iq OK" gpurun_out/r06/repro.txt; grep FAIL gpurun_out/r06/repro.txt | cut -c1-200 | head
timeout 2400 python -m pytest tests -m gpu -q -rf 2>&1 | grep -E "^FAILED|passed|failed" | cut -c1-220 > gpurun_out/r06/pytest_gpu.txt; tail -5 gpurun_out/r06/pytest_gpu.txt
S=${1:-90}
for mode in "--long" "--long --ingest8" "--long --stream" "--slice128" "" "--ingest8" "--stream"; do
  n=$(echo $mode | tr -d ' -'); n=${n:-general}
  timeout $((S + 300)) python tools/fuzz_engine.py $mode --seconds $S --seed $((1000 + ${#n})) > gpurun_out/r06/fuzz_$n.txt 2>&1; echo "fuzz $n: $(tail -1 gpurun_out/r06/fuzz_$n.txt | cut -c1-200)"
done
