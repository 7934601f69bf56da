#!/bin/bash
# Round-6 evidence run on the MI355X box, ONE gpurun call: the headline line with the driver's flags (carries group_path,
# north_star_shape, clocks, board sample) and with the defaults, the many-channel shapes, the long-filter shapes second
# generation against first, rocprofv3 kernel stats of the headline command and of the 1024-channel shape, PMC passes (SQ
# counters; FETCH_SIZE and WRITE_SIZE each on its own) for the headline, 1024 channels and the four long-filter shapes.
# Outputs under gpurun_out/r06e/; tools/collect_r06.py turns them into profiles/r06_*.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06e; rm -rf $O; mkdir -p $O
B="--no-fp32 --no-chain --no-series"
N="--no-cpu-baseline $B"
sha256sum tsl-sdr_amd/libmultifm_hip.so > $O/library.sha256
T0=$SECONDS; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driverflags.json 2> $O/bench_driverflags.err; echo "driver-flags run (everything the default line carries): $((SECONDS - T0)) s wall" > $O/driver_run_s.txt
timeout 400 python bench.py $B > $O/bench_default.json 2> $O/bench_default.err
timeout 300 python bench.py --kernel mfma1 $N > $O/bench_mfma1.json 2> $O/bench_mfma1.err
timeout 300 python bench.py --config cfg2_64ch_grid $N > $O/bench_grid64.json 2> $O/bench_grid64.err
for c in 128 256 1024; do
  timeout 600 python bench.py --config cfg3_1024ch --channels-per-gpu $c --steps 40 --warmup 5 $N > $O/bench_c$c.json 2> $O/bench_c$c.err
  timeout 600 python bench.py --config cfg3_1024ch --channels-per-gpu $c --kernel slice128 --steps 40 --warmup 5 $N > $O/bench_c${c}_slice128.json 2> $O/bench_c${c}_slice128.err
  timeout 600 python bench.py --config cfg3_1024ch --channels-per-gpu $c --kernel slice64 --steps 40 --warmup 5 $N > $O/bench_c${c}_slice64.json 2> $O/bench_c${c}_slice64.err
done
# long filters: the second-generation long-filter kernel (auto) against the first generation (mfma1, what rounds 3-4 ran)
for s in "cfg5 cfg5_airspy 256" "t512 cfg2_64ch_512taps 64" "t256 cfg2_64ch_256taps 64" "d25 pocsag_rtlsdr_256taps 64" "d100 pocsag_airspy 64" "d120 multifm_airspy 64"; do
  set -- $s
  for k in auto mfma1; do
    timeout 600 python bench.py --config $2 --channels-per-gpu $3 --kernel $k --steps 40 --warmup 5 $N > $O/bench_$1_$k.json 2> $O/bench_$1_$k.err
  done
done
timeout 600 python bench.py --config cfg5_airspy --channels-per-gpu 256 --kernel v3l1 --steps 40 --warmup 5 $N > $O/bench_cfg5_v3l1.json 2> $O/bench_cfg5_v3l1.err
timeout 600 python bench.py --config pocsag_rtlsdr --channels-per-gpu 64 --steps 60 --warmup 5 $N > $O/bench_pocsag_d25.json 2> $O/bench_pocsag_d25.err
timeout 600 python bench.py --config multifm_1ch --channels-per-gpu 64 --steps 60 --warmup 5 $N > $O/bench_multifm_d40.json 2> $O/bench_multifm_d40.err
# profiled runs: no 0.6 s of sustained load in front of the board sample (5 000 more dispatches per run in every trace and counter
# file: the call's output went over gpurun's 64 MiB), and the per-dispatch traces of the counter passes are not kept
export BENCH_BOARD_SAMPLE_AFTER_S=0
# rocprofv3 kernel trace of the headline command (same flags the driver uses) and of the 1024-channel shape
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kstats -o k -- python3 bench.py --gpus 1 --steps 20 --warmup 5 $N > $O/kstats.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kstats1024 -o k -- python3 bench.py --config cfg3_1024ch --channels-per-gpu 1024 --steps 20 --warmup 3 --settle-seconds 0.3 $N > $O/kstats1024.log 2>&1
# counters: SQ passes, then the two HBM byte counters, each alone
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
pmc head
pmc c1024 --config cfg3_1024ch --channels-per-gpu 1024
pmc c1024s128 --config cfg3_1024ch --channels-per-gpu 1024 --kernel slice128
pmc c1024s64 --config cfg3_1024ch --channels-per-gpu 1024 --kernel slice64
pmc c1024wb --config cfg3_1024ch --channels-per-gpu 1024 --pcm-write-back
pmc cfg5 --config cfg5_airspy --channels-per-gpu 256
pmc d25 --config pocsag_rtlsdr_256taps --channels-per-gpu 64
pmc d100 --config pocsag_airspy --channels-per-gpu 64
pmc d120 --config multifm_airspy --channels-per-gpu 64
unset BENCH_BOARD_SAMPLE_AFTER_S
# what the line's own instrumentation costs, the link, host-fed end to end
bash tools/r06/step_overheads.sh > $O/step_overheads.txt 2>&1
timeout 300 python3 tools/r06/link_probe.py > $O/link_probe.txt 2>&1
nproc > $O/host.txt; grep -m1 "model name" /proc/cpuinfo >> $O/host.txt
du -sh $O; ls $O | wc -l
