#!/bin/bash
# round-2 evidence for the float path: bench lines (persistent / round-1 kernel, both shapes), rocprofv3 kernel stats,
# SQ counters, the fp32-MFMA shadow microbenchmark
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/f32r02
mkdir -p $O
for r in 1 2; do
  python tools/bench_f32.py --iters 30 2>/dev/null | tail -1 >> $O/bench.jsonl
  python tools/bench_f32.py --iters 30 --tile-kernel 2>/dev/null | tail -1 >> $O/bench_tile_kernel.jsonl
done
python tools/bench_f32.py --config cfg5_airspy --channels 256 --iters 10 2>/dev/null | tail -1 >> $O/bench.jsonl
python tools/bench_f32.py --config cfg5_airspy --channels 256 --iters 10 --tile-kernel 2>/dev/null | tail -1 >> $O/bench_tile_kernel.jsonl
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kstats -o k -- python3 tools/bench_f32.py --iters 20 > $O/kstats.log 2>&1
bash tools/pmc_f32.sh > $O/pmc.txt 2>&1
./tools/ubench_shadow_f32 > $O/ubench_shadow_f32.txt 2>&1
cat $O/bench.jsonl $O/bench_tile_kernel.jsonl | cut -c1-260
find $O/kstats -name "*kernel_stats.csv" | head -1 | xargs -r cut -c1-200 | head -6
tail -22 $O/pmc.txt
