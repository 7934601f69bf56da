#!/bin/bash
# round 6: the gpu suite, first-contact scripts and fuzz of the scheduled long-filter kernel, then the slice A/B with error bars
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python tools/r05/v3l_smoke.py > gpurun_out/r06/v3l_smoke.txt 2>&1; tail -1 gpurun_out/r06/v3l_smoke.txt; grep -c FAIL gpurun_out/r06/v3l_smoke.txt
timeout 600 python tools/r06/slice128_smoke.py > gpurun_out/r06/slice128_smoke.txt 2>&1; tail -1 gpurun_out/r06/slice128_smoke.txt
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee gpurun_out/r06/pytest_gpu.txt
S=${1:-120}
for mode in "--long" "--long --ingest8" "--long --stream" "--slice128" "" "--ingest8" "--stream"; do
  n=$(echo $mode | tr -d ' -'); n=${n:-general}
  timeout $((S + 300)) python tools/fuzz_engine.py $mode --seconds $S --seed $((900 + ${#n})) > gpurun_out/r06/fuzz_$n.txt 2>&1; echo "fuzz $n: $(tail -1 gpurun_out/r06/fuzz_$n.txt | cut -c1-220)"
done
