#!/bin/bash
# Shader clock and power while bench.py's kernel runs back to back (one sample per second; the busy ones stand out by
# their power): tools/clocks_under_load.sh [steps]
STEPS=${1:-150000}
python -c "import torch" 2>/dev/null   # page the image in first
python bench.py --steps $STEPS --warmup 150 --no-cpu-baseline > /tmp/clk_bench.log 2>&1 &
BP=$!
for i in $(seq 1 45); do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | sed 's/.*: //' | tr '\n' ' '; echo
  sleep 1
  kill -0 $BP 2>/dev/null || break
done
wait $BP
tail -1 /tmp/clk_bench.log | cut -c1-200
