#!/bin/bash
# HBM bytes of the 1024-channel shape: general plan (every channel streams its rotator table) against the raster plan (no tables)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03pmc1024; rm -rf $O; mkdir -p $O
N="--no-cpu-baseline --no-fp32 --no-chain"
for cfg in cfg3_1024ch cfg3_1024ch_grid; do
  P="python3 bench.py --config $cfg --channels-per-gpu 1024 --steps 6 --warmup 2 --settle-seconds 0.3 $N"
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${cfg}_fetch -o f -- $P > $O/${cfg}_fetch.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${cfg}_write -o w -- $P > $O/${cfg}_write.log 2>&1
done
python3 - <<'PY'
import csv,glob
for cfg in ("cfg3_1024ch","cfg3_1024ch_grid"):
    r={}
    for name,d in (("FETCH_SIZE","fetch"),("WRITE_SIZE","write")):
        fs=glob.glob(f"gpurun_out/r03pmc1024/{cfg}_{d}/**/*counter_collection.csv",recursive=True)
        v=[float(x["Counter_Value"]) for x in csv.DictReader(open(fs[0])) if "channel_kernel" in x["Kernel_Name"] and x["Counter_Name"]==name]
        v=v[len(v)//2:]; r[name]=sum(v)/len(v)
    print(cfg, "FETCH_SIZE %.0f KB WRITE_SIZE %.0f KB -> %.1f MB per launch (FETCH x2 + WRITE)" % (r["FETCH_SIZE"], r["WRITE_SIZE"], (2*r["FETCH_SIZE"]+r["WRITE_SIZE"])*1024/1e6))
PY
