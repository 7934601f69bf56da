#!/bin/bash
# what the board allows and what the shapes draw: hwmon power cap, then the bench line's own clock / board sample per shape
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp
O=gpurun_out/exp/power_cap.txt
{
for h in /sys/class/drm/card*/device/hwmon/hwmon*; do
  echo "== $h"
  for f in power1_cap power1_cap_max power1_cap_min power1_cap_default power1_average power1_input power1_label freq1_input freq1_label temp1_input; do
    [ -r $h/$f ] && echo "$f $(cat $h/$f 2>/dev/null)"
  done
done
for d in /sys/class/drm/card*/device; do
  echo "== $d"; for f in pp_dpm_sclk pp_power_profile_mode power_dpm_force_performance_level; do [ -r $d/$f ] && { echo "-- $f"; head -20 $d/$f; }; done
done
rocm-smi --showpower --showclocks --showmaxpower 2>&1 | head -40
} > $O 2>&1
N="--no-cpu-baseline --no-fp32 --no-chain --no-series --steps 300 --warmup 20 --settle-seconds 1.0"
for s in "cfg2_64ch 64" "cfg2_64ch_grid 64" "cfg3_1024ch 1024" "cfg5_airspy 256" "multifm_airspy 64" "pocsag_airspy 64"; do
  set -- $s
  timeout 300 python bench.py --config $1 --channels-per-gpu $2 $N 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; c=r.get('clocks',{}); b=r.get('board_sample',{})
print('$1 $2', 'kernel_ms', round(r['kernel_ms'],4), 'cycles', c.get('shader_ticks_median'), 'MHz_in_launches', round(c.get('sclk_mhz_effective') or 0), 'board', json.dumps(b)[:300])" >> $O 2>&1
done
timeout 120 python bench.py --input rtlsdr_u8 $N 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; c=r.get('clocks',{}); b=r.get('board_sample',{})
print('cfg2 rtlsdr_u8', 'kernel_ms', round(r['kernel_ms'],4), 'cycles', c.get('shader_ticks_median'), 'MHz_in_launches', round(c.get('sclk_mhz_effective') or 0), 'board', json.dumps(b)[:300])" >> $O 2>&1
cat $O
