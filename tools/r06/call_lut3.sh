#!/bin/bash
# round 6: the asm table reads issued one by one, each behind its own division (mode 2), in all instances of the long-filter kernel
# (tools/exp/libexp_st2.so: k-step counts 4, 8, 16 rebuilt) against the shipped library (mode 1 with one row block per wave, the
# compiler's reads with two)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06lut3; mkdir -p $O
MFM_LIB=$PWD/tools/exp/libexp_st2.so timeout 900 python tools/r06/slice128_smoke.py > $O/slice128_smoke.txt 2>&1; tail -n 1 $O/slice128_smoke.txt
MFM_LIB=$PWD/tools/exp/libexp_st2.so timeout 900 python tools/r05/v3l_smoke.py > $O/v3l_smoke.txt 2>&1; tail -n 1 $O/v3l_smoke.txt
MFM_LIB=$PWD/tools/exp/libexp_st2.so timeout 900 python tools/r06/repro.py > $O/repro.txt 2>&1; tail -n 1 $O/repro.txt | cut -c1-120
for s in "c1024 --config cfg3_1024ch --channels-per-gpu 1024" "cfg5 --config cfg5_airspy --channels-per-gpu 256" "d120 --config multifm_airspy" "d25 --config pocsag_rtlsdr_256taps" "d100 --config pocsag_airspy" "t512 --config cfg2_64ch_512taps" "t256 --config cfg2_64ch_256taps"; do
  set -- $s; tag=$1; shift
  timeout 1500 python tools/exp/ab.py --reps 6 --bench-args "$*" --out $O/ab_$tag.txt "shipped=" "reads_one_by_one=tools/exp/libexp_st2.so" > $O/ab_$tag.log 2>&1; tail -n 3 $O/ab_$tag.log
done
