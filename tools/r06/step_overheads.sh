#!/bin/bash
# Round 5: what the line's new instrumentation costs the timed region (20 steps of 0.12 ms): the sysfs sample, timing every launch
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05; mkdir -p $O
F="--gpus 1 --warmup 5 --no-cpu-baseline --no-fp32 --no-chain --no-series"
show() { python3 -c "
import json,sys
l=json.loads(open('$1').read().strip().splitlines()[-1]); r=l['roofline']
print('$2: ms_per_step %.4f kernel_ms %.4f timed %d value %.3g board %s' % (l['ms_per_step'], r['kernel_ms'], r['timed_launches'], l['value'], r.get('board_sample')))"; }
python bench.py $F --steps 20 > $O/so1.json 2>/dev/null; show $O/so1.json "20 steps, sampler on"
BENCH_NO_BOARD_SAMPLE=1 python bench.py $F --steps 20 > $O/so2.json 2>/dev/null; show $O/so2.json "20 steps, sampler off"
BENCH_NO_BOARD_SAMPLE=1 python bench.py $F --steps 64 > $O/so3.json 2>/dev/null; show $O/so3.json "64 steps (sparse timing), sampler off"
python bench.py $F --steps 64 > $O/so4.json 2>/dev/null; show $O/so4.json "64 steps (sparse timing), sampler on"
BENCH_NO_BOARD_SAMPLE=1 python bench.py $F --steps 300 --warmup 150 > $O/so5.json 2>/dev/null; show $O/so5.json "300 steps, sampler off"
