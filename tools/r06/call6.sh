#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python tools/r05/v3l_smoke.py quick > gpurun_out/r06/v3l_smoke.txt 2>&1; tail -1 gpurun_out/r06/v3l_smoke.txt
timeout 500 python tools/r06/repro.py > gpurun_out/r06/repro.txt 2>&1; grep -c " OK iq OK" gpurun_out/r06/repro.txt; grep FAIL gpurun_out/r06/repro.txt | head
B="python bench.py --no-cpu-baseline --no-fp32 --no-chain --no-series --steps 200 --warmup 20 --settle-seconds 0.5"
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; c=r.get("clocks") or {}; print(round(r["kernel_ms"]*1000,1), round(d["ms_per_step"]*1000,1), "verified", d.get("verified"), "cycles", c.get("shader_ticks_median"), round(c.get("sclk_mhz_effective") or 0), r.get("kernel"))'
for rep in 1 2 3; do
echo -n "cfg5 256ch: "; timeout 300 $B --config cfg5_airspy --channels-per-gpu 256 2>/dev/null | tail -1 | python -c "$P"
echo -n "d120 512t: "; timeout 300 $B --config multifm_airspy 2>/dev/null | tail -1 | python -c "$P"
echo -n "d100 256t: "; timeout 300 $B --config pocsag_airspy 2>/dev/null | tail -1 | python -c "$P"
echo -n "d25 256t: "; timeout 300 $B --config pocsag_rtlsdr_256taps 2>/dev/null | tail -1 | python -c "$P"
done 2>&1 | tee gpurun_out/r06/call6_timing.txt
timeout 2400 python -m pytest tests -m gpu -q -rf 2>&1 | grep -E "^FAILED|passed|failed" | cut -c1-220 > gpurun_out/r06/pytest_gpu.txt; tail -3 gpurun_out/r06/pytest_gpu.txt
S=60
for mode in "--long" "--long --ingest8" "--long --stream" "--slice128" "" "--ingest8" "--stream"; do
  n=$(echo $mode | tr -d ' -'); n=${n:-general}
  timeout $((S + 300)) python tools/fuzz_engine.py $mode --seconds $S --seed $((1000 + ${#n})) > gpurun_out/r06/fuzz_$n.txt 2>&1; echo "fuzz $n: $(tail -1 gpurun_out/r06/fuzz_$n.txt | cut -c1-200)"
done
timeout 900 python tools/exp/ab.py --reps 6 --out gpurun_out/r06/ab_toeplitz.txt "toeplitz_off=tools/exp/libexp_toep0.so" "toeplitz_on=" 2>&1 | tail -5
for c in 128 256 1024; do
timeout 1500 python tools/exp/ab.py --reps 6 --bench-args "--config cfg3_1024ch --channels-per-gpu $c" --out gpurun_out/r06/ab_slice128_$c.txt "slice64=flags:--kernel slice64" "slice128=flags:--kernel slice128" 2>&1 | tail -4
done
