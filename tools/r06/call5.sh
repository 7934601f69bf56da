#!/bin/bash
# round 6: timing of the shapes the long-filter kernel runs (rotator entries in the staging registers), store-policy A/B with error bars
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python tools/r05/v3l_smoke.py quick > gpurun_out/r06/v3l_smoke.txt 2>&1; tail -1 gpurun_out/r06/v3l_smoke.txt
timeout 500 python tools/r06/repro.py 2>&1 | grep -c OK
B="python bench.py --no-cpu-baseline --no-fp32 --no-chain --no-series --steps 200 --warmup 20 --settle-seconds 0.5"
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; c=r.get("clocks") or {}; print(round(r["kernel_ms"]*1000,1), round(d["ms_per_step"]*1000,1), "verified", d.get("verified"), "cycles", c.get("shader_ticks_median"), round(c.get("sclk_mhz_effective") or 0), r.get("kernel"))'
for rep in 1 2 3; do
echo -n "cfg5 256ch: "; timeout 300 $B --config cfg5_airspy --channels-per-gpu 256 2>/dev/null | tail -1 | python -c "$P"
echo -n "d120 512t: "; timeout 300 $B --config multifm_airspy 2>/dev/null | tail -1 | python -c "$P"
echo -n "d100 256t: "; timeout 300 $B --config pocsag_airspy 2>/dev/null | tail -1 | python -c "$P"
echo -n "d25 256t: "; timeout 300 $B --config pocsag_rtlsdr_256taps 2>/dev/null | tail -1 | python -c "$P"
echo -n "t512 d96: "; timeout 300 $B --config cfg2_64ch_512taps 2>/dev/null | tail -1 | python -c "$P"
echo -n "t256 d96: "; timeout 300 $B --config cfg2_64ch_256taps 2>/dev/null | tail -1 | python -c "$P"
echo -n "headline: "; timeout 300 $B 2>/dev/null | tail -1 | python -c "$P"
done 2>&1 | tee gpurun_out/r06/call5_timing.txt
for c in 1024 256; do
timeout 1500 python tools/exp/ab.py --reps 6 --bench-args "--config cfg3_1024ch --channels-per-gpu $c" --out gpurun_out/r06/ab_store_policy_$c.txt "write_back=flags:--pcm-write-back" "write_through=flags:" 2>&1 | tail -6
done
