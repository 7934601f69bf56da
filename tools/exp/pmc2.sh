#!/bin/bash
# pmc2.sh <lib-suffix> "<counters...>" : mean per launch of the listed counters for the channel kernel
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
x=$1; shift
mkdir -p gpurun_out/exp; rm -rf gpurun_out/exp/pmc2_$x
MFM_LIB=$PWD/tools/exp/libexp_$x.so timeout 200 rocprofv3 --pmc $@ --kernel-trace --output-format csv -d gpurun_out/exp/pmc2_$x -o p -- python3 bench.py --steps 20 --warmup 5 --settle-seconds 0.3 --no-cpu-baseline --no-fp32 ${BENCH_ARGS} > gpurun_out/exp/pmc2_$x.log 2>&1
python3 - "$x" <<'PY'
import csv,glob,sys,collections
x=sys.argv[1]
acc=collections.defaultdict(list)
for fn in glob.glob(f"gpurun_out/exp/pmc2_{x}/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(fn)):
        if "channel_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(acc.items()):
    v=v[len(v)//2:]
    print("X=%s %-28s %.4g"%(x,k,sum(v)/max(1,len(v))))
rows=[r for fn in glob.glob(f"gpurun_out/exp/pmc2_{x}/**/*kernel_trace.csv",recursive=True) for r in csv.DictReader(open(fn)) if "channel_kernel" in r["Kernel_Name"]]
d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in rows]; d=d[len(d)//2:]
print("X=%s duration_us %.1f"%(x,sum(d)/len(d)))
PY
