#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06ab; mkdir -p $O
for s in "cfg5 --config cfg5_airspy --channels-per-gpu 256" "d120 --config multifm_airspy" "d100 --config pocsag_airspy" "d25 --config pocsag_rtlsdr_256taps" "t512 --config cfg2_64ch_512taps" "t256 --config cfg2_64ch_256taps" "head "; do
  set -- $s; tag=$1; shift
  timeout 1500 python tools/exp/ab.py --reps 6 --bench-args "$*" --out $O/ab_r05_$tag.txt "round5=tools/exp/libexp_r05.so" "round6=" 2>&1 | tail -3
done
for c in 512 768; do
timeout 1500 python tools/exp/ab.py --reps 6 --bench-args "--config cfg3_1024ch --channels-per-gpu $c" --out $O/ab_slice128_$c.txt "slice64=flags:--kernel slice64" "slice128=flags:--kernel slice128" 2>&1 | tail -3
done
