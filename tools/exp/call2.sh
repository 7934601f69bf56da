#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp
MFM_LIB=$PWD/tools/exp/libexp_toep.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3 > gpurun_out/exp/toep_parity.txt
cat gpurun_out/exp/toep_parity.txt
AB_REPS=4 tools/exp/run.sh base toep > gpurun_out/exp/ab2.txt 2>&1
AB_REPS=2 BENCH_ARGS="--config cfg2_64ch_grid" tools/exp/run.sh base toep >> gpurun_out/exp/ab2.txt 2>&1
AB_REPS=2 BENCH_ARGS="--config cfg3_1024ch --channels-per-gpu 1024 --steps 40" tools/exp/run.sh base toep >> gpurun_out/exp/ab2.txt 2>&1
cat gpurun_out/exp/ab2.txt
