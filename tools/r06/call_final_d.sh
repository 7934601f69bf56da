#!/bin/bash
# round 6, the tree as committed: whole gpu suite, smoke, fuzz soak (seven modes), then the evidence run (tools/prof_r06.sh)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06g; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.txt
grep -h "passed\|failed" $O/pytest_gpu.txt | tail -n 2
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" > $O/smoke.txt 2>&1; tail -n 1 $O/smoke.txt
bash tools/r06/fuzz.sh 100 6000
bash tools/prof_r06.sh > $O/prof.log 2>&1; tail -n 2 $O/prof.log
