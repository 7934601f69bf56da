#!/bin/bash
# Round-4 evidence run on the MI355X box: bench lines (headline with the driver's flags and with the defaults, the exact-grid
# plans, the many-channel shapes, the other kernels, the reference's own geometries), rocprofv3 kernel stats of the headline
# command, PMC passes (SQ counters, FETCH_SIZE, WRITE_SIZE each on its own).  Outputs under gpurun_out/r04/;
# tools/collect_r04.py turns them into profiles/r04_* and regenerates the marked sections of profiles/README.md and DESIGN.md.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; rm -rf $O; mkdir -p $O
B="--no-fp32 --no-chain --no-series"
N="--no-cpu-baseline $B"
T0=$SECONDS; timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driverflags.json 2> $O/bench_driverflags.err; echo "driver-flags run (everything the default line carries): $((SECONDS - T0)) s wall" > $O/driver_run_s.txt
timeout 400 python bench.py $B > $O/bench_default.json 2> $O/bench_default.err
timeout 300 python bench.py --overlap $N > $O/bench_overlap.json 2> $O/bench_overlap.err
timeout 300 python bench.py --kernel mfma1 $N > $O/bench_mfma1.json 2> $O/bench_mfma1.err
timeout 300 python bench.py --kernel dot2 $N --steps 60 --warmup 10 > $O/bench_dot2.json 2> $O/bench_dot2.err
timeout 300 python bench.py --config cfg2_64ch_grid $N > $O/bench_grid64.json 2> $O/bench_grid64.err
for c in 128 256 1024; do
  timeout 600 python bench.py --config cfg3_1024ch --channels-per-gpu $c --steps 40 --warmup 5 $N > $O/bench_c$c.json 2> $O/bench_c$c.err
done
timeout 600 python bench.py --config cfg3_1024ch_grid --channels-per-gpu 1024 --steps 40 --warmup 5 $N > $O/bench_grid1024.json 2> $O/bench_grid1024.err
timeout 600 python bench.py --config cfg5_airspy --channels-per-gpu 256 --steps 40 --warmup 5 $N > $O/bench_cfg5_256.json 2> $O/bench_cfg5_256.err
timeout 600 python bench.py --config cfg5_airspy --channels-per-gpu 256 --kernel mfma1s --steps 40 --warmup 5 $N > $O/bench_cfg5_256_streamed.json 2> $O/bench_cfg5_256_streamed.err
for t in 512 256; do for k in auto mfma1s; do
  timeout 600 python bench.py --config cfg2_64ch_${t}taps --kernel $k --steps 40 --warmup 5 $N > $O/bench_t${t}_$k.json 2> $O/bench_t${t}_$k.err
done; done
timeout 600 python bench.py --config pocsag_rtlsdr --channels-per-gpu 64 --steps 60 --warmup 5 $N > $O/bench_pocsag_d25.json 2> $O/bench_pocsag_d25.err
timeout 600 python bench.py --config multifm_1ch --channels-per-gpu 64 --steps 60 --warmup 5 $N > $O/bench_multifm_d40.json 2> $O/bench_multifm_d40.err
timeout 600 python bench.py --config multifm_1ch --channels-per-gpu 64 --kernel mfma1 --steps 60 --warmup 5 $N > $O/bench_multifm_d40_mfma1.json 2> $O/bench_multifm_d40_mfma1.err
# rocprofv3 kernel trace of the headline command (same flags the driver uses)
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kstats -o k -- python3 bench.py --gpus 1 --steps 20 --warmup 5 $N > $O/kstats.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kstats1024 -o k -- python3 bench.py --config cfg3_1024ch --channels-per-gpu 1024 --steps 20 --warmup 3 --settle-seconds 0.3 $N > $O/kstats1024.log 2>&1
# counters: SQ passes, then the two HBM byte counters, each alone
P="python3 bench.py --steps 8 --warmup 3 --settle-seconds 0.3 $N"
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/p1 -o p -- $P > $O/p1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES --kernel-trace --output-format csv -d $O/p2 -o p -- $P > $O/p2.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/p3 -o p -- $P > $O/p3.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o f -- $P > $O/fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o w -- $P > $O/write.log 2>&1
P1024="python3 bench.py --config cfg3_1024ch --channels-per-gpu 1024 --steps 6 --warmup 2 --settle-seconds 0.3 $N"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch1024 -o f -- $P1024 > $O/fetch1024.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write1024 -o w -- $P1024 > $O/write1024.log 2>&1
# known-byte kernels for the two byte counters (tools/ubench_hbm.hip: streaming 16-byte reads, 8-byte stores of 128 bytes per row)
timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/cal_fetch -o f -- tools/ubench_hbm calib > $O/cal_fetch.log 2>&1
timeout 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/cal_write -o w -- tools/ubench_hbm calib > $O/cal_write.log 2>&1
nproc > $O/host.txt; grep -m1 "model name" /proc/cpuinfo >> $O/host.txt
ls $O | head -60
