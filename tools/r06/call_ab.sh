#!/bin/bash
# round 6: the A/Bs that go into profiles/ (tools/exp/ab.py: six alternating repetitions, 2-standard-error rule), one box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06ab; mkdir -p $O
timeout 900 python tools/exp/ab.py --reps 6 --out $O/ab_toeplitz.txt "toeplitz_off=tools/exp/libexp_toep0.so" "toeplitz_on=" 2>&1 | tail -4
for c in 128 256 1024; do
timeout 1500 python tools/exp/ab.py --reps 6 --bench-args "--config cfg3_1024ch --channels-per-gpu $c" --out $O/ab_slice128_$c.txt "slice64=flags:--kernel slice64" "slice128=flags:--kernel slice128" 2>&1 | tail -3
done
# round 6's library against round 5's (commit 16e5fa1 built as it was), shape by shape
for s in "cfg5 --config cfg5_airspy --channels-per-gpu 256" "d120 --config multifm_airspy" "d100 --config pocsag_airspy" "d25 --config pocsag_rtlsdr_256taps" "t512 --config cfg2_64ch_512taps" "t256 --config cfg2_64ch_256taps" "head "; do
  set -- $s; tag=$1; shift
  timeout 1500 python tools/exp/ab.py --reps 6 --bench-args "$*" --out $O/ab_r05_$tag.txt "round5=tools/exp/libexp_r05.so" "round6=" 2>&1 | tail -3
done
for c in 1024 256 64; do
timeout 1500 python tools/exp/ab.py --reps 6 --bench-args "--config cfg3_1024ch --channels-per-gpu $c" --out $O/ab_store_policy_$c.txt "write_back=flags:--pcm-write-back" "default=flags:" 2>&1 | tail -3
done
