#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 500 python tools/r06/repro.py 2>&1 | tail -30
timeout 600 python tools/fuzz_engine.py --long --ingest8 --seconds 60 --seed 911 2>&1 | tail -1 | cut -c1-400
timeout 600 python tools/fuzz_engine.py --slice128 --seconds 60 --seed 908 2>&1 | tail -1 | cut -c1-400
timeout 600 python tools/fuzz_engine.py --stream --seconds 100 --seed 906 2>&1 | tail -1 | cut -c1-400
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee gpurun_out/r06/pytest_gpu.txt
