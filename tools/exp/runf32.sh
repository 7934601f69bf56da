#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do for x in "$@"; do echo -n "X=$x: "; MFM_LIB=$PWD/tools/exp/libexp_$x.so timeout 200 python tools/bench_f32.py --iters 30 2>/dev/null | head -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_block'], d['fp32_tflops'])"; done; done
