#!/bin/bash
# round 3: the whole GPU suite, then randomised GPU-vs-oracle campaigns on the round's final kernels
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03fuzz; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
{
echo "python tools/fuzz_engine.py --seconds 600 --seed 301"; timeout 900 python tools/fuzz_engine.py --seconds 600 --seed 301 2>&1 | tail -3
echo "python tools/fuzz_engine.py --seconds 300 --seed 302"; timeout 600 python tools/fuzz_engine.py --seconds 300 --seed 302 2>&1 | tail -3
echo "python tools/fuzz_engine.py --ingest8 --seconds 240 --seed 303"; timeout 500 python tools/fuzz_engine.py --ingest8 --seconds 240 --seed 303 2>&1 | tail -3
echo "python tools/fuzz_stages.py --seconds 200 --seed 304"; timeout 500 python tools/fuzz_stages.py --seconds 200 --seed 304 2>&1 | tail -3
} > $O/fuzz.log 2>&1
cat $O/fuzz.log
