#!/bin/bash
# validation of the build with the Toeplitz re-use of B fragments: the whole gpu suite, a short fuzz of the modes that reach the
# D = 96 instances (general, stream, 8-bit), and the 8-bit line against the int16 one
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3
timeout 300 python tools/fuzz_engine.py --seconds 100 --seed 811 2>&1 | tail -1 | cut -c1-200
timeout 300 python tools/fuzz_engine.py --stream --seconds 100 --seed 812 2>&1 | tail -1 | cut -c1-200
timeout 300 python tools/fuzz_engine.py --ingest8 --seconds 100 --seed 813 2>&1 | tail -1 | cut -c1-200
for i in 1 2; do
timeout 200 python bench.py --no-cpu-baseline --no-fp32 --no-chain --no-series --steps 200 --warmup 20 --settle-seconds 0.5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('int16', round(r['kernel_ms']*1000,1), d['verified'], 'in8', d['ingest_8bit']['kernel_ms'])"
done
