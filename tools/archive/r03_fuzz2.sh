#!/bin/bash
# round 3, second campaign on the final build: longer runs, other seeds
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03fuzz2; rm -rf $O; mkdir -p $O
{
echo "python tools/fuzz_engine.py --seconds 900 --seed 311"; timeout 1200 python tools/fuzz_engine.py --seconds 900 --seed 311 2>&1 | tail -2
echo "python tools/fuzz_engine.py --ingest8 --seconds 420 --seed 312"; timeout 700 python tools/fuzz_engine.py --ingest8 --seconds 420 --seed 312 2>&1 | tail -2
echo "python tools/fuzz_stages.py --seconds 300 --seed 313"; timeout 600 python tools/fuzz_stages.py --seconds 300 --seed 313 2>&1 | tail -2
} > $O/fuzz.log 2>&1
cat $O/fuzz.log
