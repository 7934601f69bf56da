#!/bin/bash
# clock and counters with one workgroup per CU against two (tools/archive/r04_call16.sh has the times)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04p; rm -rf $O; mkdir -p $O
B="--no-cpu-baseline --no-fp32 --no-chain --no-series"
P="python3 bench.py --steps 8 --warmup 3 --settle-seconds 0.5 $B"
for v in base onewg; do
  L=""; [ $v != base ] && L=$PWD/tools/exp/libexp_$v.so
  export MFM_LIB=$L
  timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc_$v -o p -- $P > $O/pmc_$v.log 2>&1
  python3 - $O/pmc_$v $v <<'PY'
import csv, glob, sys, collections
d, v = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
dur = []
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mfm_channel_kernel_v3" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mfm_channel_kernel_v3" in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = {k: sum(x[len(x)//2:]) / max(1, len(x[len(x)//2:])) for k, x in acc.items()}
du = sum(dur[len(dur)//2:]) / max(1, len(dur[len(dur)//2:]))
print(v, "duration us %.1f" % du, {k: "%.4g" % a for k, a in sorted(out.items())})
if "GRBM_GUI_ACTIVE" in out:
    cyc = out["GRBM_GUI_ACTIVE"] / 8
    print(v, "cycles per launch %.0f -> clock %.3f GHz (serialized profiling run)" % (cyc, cyc / du / 1e3))
PY
done
unset MFM_LIB
