#!/bin/bash
# round 6: what the evidence call (tools/prof_r06.sh as it was then) did not have - north star's shape on 64-channel slices, now that
# the default takes 128-channel slices from 512 channels on: bench line and PMC passes into the same gpurun_out/r06e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06e; mkdir -p $O
B="--no-fp32 --no-chain --no-series"; N="--no-cpu-baseline $B"
for c in 128 256 1024; do
  timeout 600 python bench.py --config cfg3_1024ch --channels-per-gpu $c --kernel slice64 --steps 40 --warmup 5 $N > $O/bench_c${c}_slice64.json 2> $O/bench_c${c}_slice64.err
done
export BENCH_BOARD_SAMPLE_AFTER_S=0
pmc() { # tag, bench flags...
  local tag=$1; shift
  local P="python3 bench.py --steps 8 --warmup 3 --settle-seconds 0.3 $N $*"
  timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/${tag}_p1 -o p -- $P > $O/${tag}_p1.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_MFMA --kernel-trace --output-format csv -d $O/${tag}_p2 -o p -- $P > $O/${tag}_p2.log 2>&1
  timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT --kernel-trace --output-format csv -d $O/${tag}_p3 -o p -- $P > $O/${tag}_p3.log 2>&1
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${tag}_fetch -o f -- $P > $O/${tag}_fetch.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${tag}_write -o w -- $P > $O/${tag}_write.log 2>&1
  find $O/${tag}_p1 $O/${tag}_p2 $O/${tag}_p3 $O/${tag}_fetch $O/${tag}_write -name "*kernel_trace.csv" -delete 2>/dev/null
}
pmc c1024s64 --config cfg3_1024ch --channels-per-gpu 1024 --kernel slice64
du -sh $O
