#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06ab; mkdir -p $O
for c in 130 192 320 768; do
timeout 1500 python tools/exp/ab.py --reps 4 --bench-args "--config cfg3_1024ch --channels-per-gpu $c" --out $O/ab_chunking_$c.txt "chunks_any=tools/exp/libexp_pre.so" "chunks_x8=" 2>&1 | tail -3
done
timeout 1500 python tools/exp/ab.py --reps 4 --bench-args "--config cfg5_airspy --channels-per-gpu 320" --out $O/ab_chunking_cfg5_320.txt "chunks_any=tools/exp/libexp_pre.so" "chunks_x8=" 2>&1 | tail -3
for c in 768; do
timeout 1500 python tools/exp/ab.py --reps 6 --bench-args "--config cfg3_1024ch --channels-per-gpu $c" --out $O/ab_slice128_$c.txt "slice64=flags:--kernel slice64" "slice128=flags:--kernel slice128" 2>&1 | tail -3
done
