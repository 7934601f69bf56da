#!/bin/bash
# Device-resident chains (engine -> resampler -> pager stage): per-stage times and rocprofv3 kernel stats of the FLEX chain
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/chainflex
rm -rf $O; mkdir -p $O
python tools/bench_chain_dev.py > $O/bench.jsonl 2>/dev/null
python tools/bench_chain_dev.py --resampler-dot2 >> $O/bench.jsonl 2>/dev/null
python tools/bench_chain_dev.py --channels 1024 --iters 8 >> $O/bench.jsonl 2>/dev/null
python tools/bench_chain_dev.py --channels 1024 --iters 8 --resampler-dot2 >> $O/bench.jsonl 2>/dev/null
python tools/bench_chain_dev.py --dc-block --iters 8 >> $O/bench.jsonl 2>/dev/null
python tools/bench_chain_dev.py --proto pocsag >> $O/bench.jsonl 2>/dev/null
python tools/bench_chain_dev.py --proto pocsag --resampler-dot2 >> $O/bench.jsonl 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/k -o k -- python3 tools/bench_chain_dev.py --iters 20 > $O/k.log 2>&1
cat $O/bench.jsonl | cut -c1-420
find $O/k -name "*kernel_stats.csv" | head -1 | xargs -r cut -c1-160 | head -8
