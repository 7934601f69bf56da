#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 2400 python -m pytest tests -m gpu -q -rf 2>&1 | grep -E "^FAILED|passed|failed" | cut -c1-220 > gpurun_out/r06/pytest_gpu.txt; tail -3 gpurun_out/r06/pytest_gpu.txt
timeout 600 python tools/r06/slice128_smoke.py > gpurun_out/r06/slice128_smoke.txt 2>&1; tail -1 gpurun_out/r06/slice128_smoke.txt
timeout 400 python tools/fuzz_engine.py --slice128 --seconds 60 --seed 2001 2>&1 | tail -1 | cut -c1-200
bash tools/prof_r06.sh 2>&1 | tail -5
