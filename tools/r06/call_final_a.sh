#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 2400 python -m pytest tests -m gpu -q -rf 2>&1 | grep -E "^FAILED|passed|failed" | cut -c1-220 > gpurun_out/r06/pytest_gpu.txt; tail -3 gpurun_out/r06/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python tools/r05/v3l_smoke.py > gpurun_out/r06/v3l_smoke.txt 2>&1; tail -1 gpurun_out/r06/v3l_smoke.txt
timeout 600 python tools/r06/slice128_smoke.py > gpurun_out/r06/slice128_smoke.txt 2>&1; tail -1 gpurun_out/r06/slice128_smoke.txt
timeout 500 python tools/r06/repro.py > gpurun_out/r06/repro.txt 2>&1; grep -c " OK iq OK" gpurun_out/r06/repro.txt; grep -c FAIL gpurun_out/r06/repro.txt
bash tools/r06/fuzz.sh 150 3000
