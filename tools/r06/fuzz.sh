#!/bin/bash
# Round 6: the fuzzers on the round's final build - filters of 129..512 taps (int16, 8-bit blocks, the coalescing / two-stream / seek
# logic), 128-tap filters on 128-channel slices, the general mix, 8-bit ingest, the stream logic - all against the oracle.
#   tools/r06/fuzz.sh [seconds per mode] [seed base]
S=${1:-150}
B=${2:-3000}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06/fuzz_final; mkdir -p $O
sha256sum tsl-sdr_amd/libmultifm_hip.so > $O/library.sha256
i=0
for mode in "--long" "--long --ingest8" "--long --stream" "--slice128" "" "--ingest8" "--stream"; do
  i=$((i + 1)); n=$(echo $mode | tr -d ' -'); n=${n:-general}
  timeout $((S + 300)) python tools/fuzz_engine.py $mode --seconds $S --seed $((B + i)) > $O/$n.txt 2>&1
  echo "== tools/fuzz_engine.py $mode --seconds $S --seed $((B + i)) ==" >> $O/all.txt; tail -1 $O/$n.txt >> $O/all.txt
  echo "fuzz $n: $(tail -1 $O/$n.txt | cut -c1-160)"
done
