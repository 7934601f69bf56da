#!/bin/bash
# round 4: the N > 1 plumbing on a one-GPU box - the RCCL path forced with one rank, `--gpus 2` where there is one GPU (must
# fail fast and loudly, not hang), and the list of counters this rocprofv3 knows.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04n; rm -rf $O; mkdir -p $O
B="--no-cpu-baseline --no-fp32 --no-chain --no-series"
BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 20 --warmup 5 $B > $O/forcedist.json 2> $O/forcedist.err; echo rc=$?
python3 -c "
import json; d=json.loads(open('$O/forcedist.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('exchange'), d['verified'])"
BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 20 --warmup 5 --input rtlsdr_u8 $B > $O/forcedist8.json 2> $O/forcedist8.err; echo rc=$?
python3 -c "
import json; d=json.loads(open('$O/forcedist8.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('exchange'), d['verified'])"
timeout 200 python bench.py --gpus 2 --steps 5 --warmup 2 > $O/gpus2.json 2> $O/gpus2.err; echo rc2=$?; grep -v "^\s*$" $O/gpus2.err | tail -4
rocprofv3 -L > $O/counters_list.txt 2>&1; grep -c . $O/counters_list.txt
