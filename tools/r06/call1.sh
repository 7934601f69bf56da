#!/bin/bash
# round 6, first contact of the scheduled long-filter kernel: correctness against the oracle, then timing of the shapes it changes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python tools/r05/v3l_smoke.py quick > gpurun_out/r06/v3l_smoke.txt 2>&1; tail -3 gpurun_out/r06/v3l_smoke.txt; grep -c FAIL gpurun_out/r06/v3l_smoke.txt
timeout 900 python tools/r06/slice128_smoke.py > gpurun_out/r06/slice128_smoke.txt 2>&1; tail -3 gpurun_out/r06/slice128_smoke.txt; grep FAIL gpurun_out/r06/slice128_smoke.txt | head
B="python bench.py --no-cpu-baseline --no-fp32 --no-chain --no-series --steps 200 --warmup 20 --settle-seconds 0.5"
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; c=r.get("clocks") or {}; print(round(r["kernel_ms"]*1000,1), round(d["ms_per_step"]*1000,1), "verified", d.get("verified"), "cycles", c.get("shader_ticks_median"), round(c.get("sclk_mhz_effective") or 0), r.get("kernel"))'
for rep in 1 2; do
echo -n "cfg5 256ch: "; timeout 300 $B --config cfg5_airspy --channels-per-gpu 256 2>/dev/null | tail -1 | python -c "$P"
echo -n "d120 512t: "; timeout 300 $B --config multifm_airspy 2>/dev/null | tail -1 | python -c "$P"
echo -n "d100 256t: "; timeout 300 $B --config pocsag_airspy 2>/dev/null | tail -1 | python -c "$P"
echo -n "d25 256t: "; timeout 300 $B --config pocsag_rtlsdr_256taps 2>/dev/null | tail -1 | python -c "$P"
for k in slice64 slice128; do for c in 128 256 1024; do
echo -n "cfg3 $c ch $k: "; timeout 300 $B --config cfg3_1024ch --channels-per-gpu $c --kernel $k 2>/dev/null | tail -1 | python -c "$P"
done; done
done 2>&1 | tee gpurun_out/r06/call1_timing.txt
