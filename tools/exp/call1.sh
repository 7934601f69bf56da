#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp
timeout 60 tools/exp/hwid_probe > gpurun_out/exp/hwid_probe.txt 2>&1
AB_REPS=3 tools/exp/run.sh base spread genprio selx > gpurun_out/exp/ab1.txt 2>&1
AB_REPS=2 BENCH_ARGS="--config cfg2_64ch_grid" tools/exp/run.sh base spread selx >> gpurun_out/exp/ab1.txt 2>&1
cat gpurun_out/exp/hwid_probe.txt; cat gpurun_out/exp/ab1.txt
