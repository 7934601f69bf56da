#!/bin/bash
# long soak of the fuzzers on the final build
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03fuzz5; mkdir -p $O
timeout 1000 python tools/fuzz_engine.py --seconds 840 --seed 51 > $O/general.txt 2>&1; tail -1 $O/general.txt
timeout 700 python tools/fuzz_engine.py --long --seconds 540 --seed 52 > $O/long.txt 2>&1; tail -1 $O/long.txt
timeout 500 python tools/fuzz_stages.py --seconds 300 --seed 53 > $O/stages.txt 2>&1; tail -1 $O/stages.txt
