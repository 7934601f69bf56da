#!/bin/bash
# round 6: the discriminator's table reads as asm statements in the long-filter kernel (MFM3_LUT_ASM): correctness, then the A/B
# against the library as it was (tools/exp/libexp_base.so), shape by shape - and the same form in the 64-channel kernel
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06lut; mkdir -p $O
timeout 900 python tools/r06/slice128_smoke.py > $O/slice128_smoke.txt 2>&1; tail -1 $O/slice128_smoke.txt
timeout 900 python tools/r05/v3l_smoke.py > $O/v3l_smoke.txt 2>&1; tail -1 $O/v3l_smoke.txt
timeout 900 python tools/r06/repro.py > $O/repro.txt 2>&1; tail -1 $O/repro.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/pytest_parity.txt 2>&1; tail -1 $O/pytest_parity.txt
for s in "cfg5 --config cfg5_airspy --channels-per-gpu 256" "c1024 --config cfg3_1024ch --channels-per-gpu 1024" "d120 --config multifm_airspy" "d25 --config pocsag_rtlsdr_256taps" "d100 --config pocsag_airspy" "t512 --config cfg2_64ch_512taps"; do
  set -- $s; tag=$1; shift
  timeout 1500 python tools/exp/ab.py --reps 6 --bench-args "$*" --out $O/ab_lut_$tag.txt "before=tools/exp/libexp_base.so" "asm_reads=" 2>&1 | tail -3
done
if [ -f tools/exp/libexp_v3lut.so ]; then
  timeout 1500 python tools/exp/ab.py --reps 6 --out $O/ab_lut_head.txt "before=" "asm_reads_v3=tools/exp/libexp_v3lut.so" 2>&1 | tail -3
  timeout 1500 python tools/exp/ab.py --reps 6 --bench-args "--config cfg3_1024ch --channels-per-gpu 256" --out $O/ab_lut_c256.txt "before=" "asm_reads_v3=tools/exp/libexp_v3lut.so" 2>&1 | tail -3
fi
for mode in "--long" "--long --ingest8" "--slice128"; do
  n=$(echo $mode | tr -d ' -')
  timeout 400 python tools/fuzz_engine.py $mode --seconds 90 --seed 4100 > $O/fuzz_$n.txt 2>&1; echo "fuzz $n: $(tail -1 $O/fuzz_$n.txt | cut -c1-160)"
done
