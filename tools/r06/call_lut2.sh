#!/bin/bash
# round 6: (1) asm table reads in the one-row-block instances only (libexp_lutrb1.so) against the library as it was (libexp_base.so);
# (2) on top of it, the epilogue's transposed rows requested one block ahead (the shipped library of this call = libexp_tpp1.so)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06lut2; mkdir -p $O
timeout 900 python tools/r06/slice128_smoke.py > $O/slice128_smoke.txt 2>&1; tail -n 1 $O/slice128_smoke.txt
timeout 900 python tools/r05/v3l_smoke.py > $O/v3l_smoke.txt 2>&1; tail -n 1 $O/v3l_smoke.txt
timeout 900 python tools/r06/repro.py > $O/repro.txt 2>&1; tail -n 1 $O/repro.txt | cut -c1-120
for s in "cfg5 --config cfg5_airspy --channels-per-gpu 256" "c1024 --config cfg3_1024ch --channels-per-gpu 1024" "d120 --config multifm_airspy" "d25 --config pocsag_rtlsdr_256taps" "d100 --config pocsag_airspy" "t512 --config cfg2_64ch_512taps" "t256 --config cfg2_64ch_256taps"; do
  set -- $s; tag=$1; shift
  timeout 1500 python tools/exp/ab.py --reps 6 --bench-args "$*" --out $O/ab_$tag.txt "before=tools/exp/libexp_base.so" "asm_reads_rb1=tools/exp/libexp_lutrb1.so" "plus_row_prefetch=tools/exp/libexp_tpp1.so" > $O/ab_$tag.log 2>&1; tail -n 4 $O/ab_$tag.log
done
for mode in "--long" "--long --ingest8" "--slice128"; do
  n=$(echo $mode | tr -d ' -')
  timeout 400 python tools/fuzz_engine.py $mode --seconds 60 --seed 4200 > $O/fuzz_$n.txt 2>&1; echo "fuzz $n: $(tail -n 1 $O/fuzz_$n.txt | cut -c1-100)"
done
