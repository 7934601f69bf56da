#!/bin/bash
# FLEX stage timing + rocprofv3 kernel stats (run through gpurun from the repo root)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out/flex
python tools/bench_pager.py --proto flex --channels 64 --samples 447392 > gpurun_out/flex/bench64.jsonl 2> gpurun_out/flex/bench64.err
python tools/bench_pager.py --proto flex --channels 1024 --samples 447392 --iters 5 > gpurun_out/flex/bench1024.jsonl 2> gpurun_out/flex/bench1024.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/flex/prof -o flex -- python3 tools/bench_pager.py --proto flex --channels 64 --samples 447392 --iters 10 > gpurun_out/flex/prof.log 2>&1
cat gpurun_out/flex/bench64.jsonl gpurun_out/flex/bench1024.jsonl
find gpurun_out/flex/prof -name "*kernel_stats.csv" | head -1 | xargs -r cut -c1-220 | head -12
