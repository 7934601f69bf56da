#!/bin/bash
# Round-2 evidence run on the MI355X box: bench lines (headline with the driver's flags and with the defaults, the
# many-channel shapes, both other kernels), rocprofv3 kernel stats of the headline command, PMC passes (SQ counters,
# FETCH_SIZE, WRITE_SIZE each on its own).  Outputs under gpurun_out/r02/; tools/collect_r02.py turns them into profiles/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02; rm -rf $O; mkdir -p $O
B="--no-fp32 --no-chain"
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driverflags.json 2> $O/bench_driverflags.err
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err
for k in mfma1 dot2; do
  timeout 300 python bench.py --kernel $k --no-cpu-baseline $B > $O/bench_$k.json 2> $O/bench_$k.err
done
for c in 128 256 1024; do
  timeout 600 python bench.py --config cfg3_1024ch --channels-per-gpu $c --steps 40 --warmup 5 --no-cpu-baseline $B > $O/bench_c$c.json 2> $O/bench_c$c.err
done
timeout 600 python bench.py --config cfg5_airspy --channels-per-gpu 256 --steps 40 --warmup 5 --no-cpu-baseline $B > $O/bench_cfg5_256.json 2> $O/bench_cfg5_256.err
timeout 600 python bench.py --config pocsag_rtlsdr --channels-per-gpu 64 --steps 60 --warmup 5 --no-cpu-baseline $B > $O/bench_pocsag_d25.json 2> $O/bench_pocsag_d25.err
# rocprofv3 kernel trace of the headline command (same flags the driver uses)
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kstats -o k -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline $B > $O/kstats.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kstats1024 -o k -- python3 bench.py --config cfg3_1024ch --channels-per-gpu 1024 --steps 20 --warmup 3 --settle-seconds 0.3 --no-cpu-baseline $B > $O/kstats1024.log 2>&1
# counters: SQ passes, then the two HBM byte counters, each alone
P="python3 bench.py --steps 8 --warmup 3 --settle-seconds 0.3 --no-cpu-baseline $B"
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/p1 -o p -- $P > $O/p1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES --kernel-trace --output-format csv -d $O/p2 -o p -- $P > $O/p2.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/p3 -o p -- $P > $O/p3.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o f -- $P > $O/fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o w -- $P > $O/write.log 2>&1
# first-generation kernel, same SQ pass, for the comparison table
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/p1_v1 -o p -- $P --kernel mfma1 > $O/p1_v1.log 2>&1
ls $O | head -40
for f in $O/bench_*.json; do echo -n "$f: "; python3 - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print(r["kernel"], "%.4g"%d["value"], "ms/step %.4f"%d["ms_per_step"], "kernel %.4f (min %.4f med %.4f p95 %.4f)"%(r["kernel_ms"], r["kernel_ms_min"], r["kernel_ms_median"], r["kernel_ms_p95"]), "frac %.3f"%r["frac"], "compute %.3f"%d["compute_roofline"]["frac"])
except Exception as e:
    print("ERR", e)
PY
done
