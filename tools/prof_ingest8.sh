#!/bin/bash
# 8-bit ingest: kernel time of the channel engine with the block read as bytes / widened first / as int16, 64 and
# 1024 channels, + rocprofv3 kernel stats of the bytes run and the widened run (run through gpurun from the repo root)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/ingest8
mkdir -p $O
: > $O/bench.jsonl
for a in "--fmt 0" "--fmt 3" "--fmt 3 --widen" "--fmt 1" "--fmt 2" "--fmt 0 --channels 1024 --steps 6" "--fmt 3 --channels 1024 --steps 6"; do
  python tools/bench_ingest8.py $a >> $O/bench.jsonl 2>> $O/bench.err
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bytes -o b -- python3 tools/bench_ingest8.py --fmt 3 > $O/prof_bytes.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_widen -o w -- python3 tools/bench_ingest8.py --fmt 3 --widen > $O/prof_widen.log 2>&1
cat $O/bench.jsonl
for d in prof_bytes prof_widen; do echo "== $d"; find $O/$d -name "*kernel_stats.csv" | head -1 | xargs -r cut -c1-200 | head -6; done
