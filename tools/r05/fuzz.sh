#!/bin/bash
# Round 5: the fuzzers on the build with the long-filter kernel (mfm_kernel_v3l.hip): filters of 129..512 taps (int16, 8-bit
# blocks, the coalescing / two-stream / seek logic) and the general mix, all against the oracle.  tools/r05/fuzz.sh [seconds per mode] [seed base, default 500]
S=${1:-240}
B=${2:-500}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05/fuzz; mkdir -p $O
timeout $((S + 200)) python tools/fuzz_engine.py --long --seconds $S --seed $((B + 1)) > $O/long.txt 2>&1; tail -1 $O/long.txt
timeout $((S + 200)) python tools/fuzz_engine.py --long --ingest8 --seconds $S --seed $((B + 2)) > $O/long8.txt 2>&1; tail -1 $O/long8.txt
timeout $((S + 200)) python tools/fuzz_engine.py --long --stream --seconds $S --seed $((B + 3)) > $O/long_stream.txt 2>&1; tail -1 $O/long_stream.txt
timeout $((S + 200)) python tools/fuzz_engine.py --seconds $S --seed $((B + 4)) > $O/general.txt 2>&1; tail -1 $O/general.txt
timeout $((S + 200)) python tools/fuzz_engine.py --stream --seconds $S --seed $((B + 5)) > $O/stream.txt 2>&1; tail -1 $O/stream.txt
