#!/bin/bash
# Round 5, second session: the driver's command on the round's final bench.py (board sample matched by PCI address) and the
# rocprofv3 kernel trace of the same command, ONE box, one gpurun call.  Outputs under gpurun_out/r05b/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05b; rm -rf $O; mkdir -p $O
T0=$SECONDS; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driverflags.json 2> $O/bench_driverflags.err; echo "driver-flags run: $((SECONDS - T0)) s wall" > $O/driver_run_s.txt
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kstats -o k -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-fp32 --no-chain --no-series > $O/kstats.log 2>&1
timeout 600 python -m pytest tests -m gpu -x -q > $O/gputest.txt 2>&1; grep -E "passed|failed|error" $O/gputest.txt | tail -3
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
find $O/kstats -name "*kernel_stats.csv" | head; cat $O/driver_run_s.txt
