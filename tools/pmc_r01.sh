#!/bin/bash
# SQ / LDS counters of the channel kernel, two passes (no other trace domains together with --pmc)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc
timeout 240 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc/p1 -o p -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fp32 > gpurun_out/pmc/p1.log 2>&1
timeout 240 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d gpurun_out/pmc/p2 -o p -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fp32 > gpurun_out/pmc/p2.log 2>&1
timeout 240 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU --kernel-trace --output-format csv -d gpurun_out/pmc/p3 -o p -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fp32 > gpurun_out/pmc/p3.log 2>&1
ls gpurun_out/pmc/*
