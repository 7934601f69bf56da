#!/bin/bash
# fuzz campaigns on the long-filter instances (resident and streamed taps), int16 and 8-bit, plus a general pass
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03fuzz3; mkdir -p $O
timeout 500 python tools/fuzz_engine.py --long --seconds 300 --seed 31 > $O/long.txt 2>&1; tail -1 $O/long.txt
timeout 400 python tools/fuzz_engine.py --long --ingest8 --seconds 240 --seed 32 > $O/long8.txt 2>&1; tail -1 $O/long8.txt
timeout 300 python tools/fuzz_engine.py --seconds 150 --seed 33 > $O/general.txt 2>&1; tail -1 $O/general.txt
