#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 2400 python -m pytest tests -m gpu -q -rf 2>&1 | grep -E "^FAILED|passed|failed" | cut -c1-220 > gpurun_out/r06/pytest_gpu.txt; tail -3 gpurun_out/r06/pytest_gpu.txt
export AB_REPS=2 BENCH_ARGS="--config cfg5_airspy --channels-per-gpu 256"
bash tools/exp/run.sh base k4 k8 k32 k64 k96 k128 2>&1 | tee gpurun_out/r06/knock2_cfg5.txt
export BENCH_ARGS="--config cfg3_1024ch --channels-per-gpu 1024 --kernel slice128"
bash tools/exp/run.sh s128base s128free 2>&1 | tee gpurun_out/r06/epi_free_1024.txt
