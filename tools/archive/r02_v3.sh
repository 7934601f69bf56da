#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "${PYTEST_K:-cfg2_64 or golden or reference_shaped or odd_geometries or channel_counts}" 2>&1 | tail -15
timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-fp32 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('bench', r['kernel'], d['value'], d['ms_per_step'], r['kernel_ms'], r['kernel_ms_min'], r['frac'])"
