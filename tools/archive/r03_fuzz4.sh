#!/bin/bash
# closing campaign on the round's final build: long filters (resident / streamed), 8-bit long filters, general engine, stages
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03fuzz4; mkdir -p $O
timeout 700 python tools/fuzz_engine.py --long --seconds 480 --seed 41 > $O/long.txt 2>&1; tail -1 $O/long.txt
timeout 500 python tools/fuzz_engine.py --long --ingest8 --seconds 300 --seed 42 > $O/long8.txt 2>&1; tail -1 $O/long8.txt
timeout 700 python tools/fuzz_engine.py --seconds 480 --seed 43 > $O/general.txt 2>&1; tail -1 $O/general.txt
timeout 400 python tools/fuzz_engine.py --ingest8 --seconds 240 --seed 44 > $O/ingest8.txt 2>&1; tail -1 $O/ingest8.txt
