#!/bin/bash
# round 4, sixth GPU call: non-temporal hints on the streaming accesses (image loads, PCM stores) - time and L2-miss traffic at
# 1024 channels (the rotator tables are what the streams push out of L2) and at the headline shape; end_to_end with lazy tickets
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04f; rm -rf $O; mkdir -p $O
summ() { python3 - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print(sys.argv[1].split('/')[-1], "%.4g"%d["value"], "ms/step %.4f"%d["ms_per_step"], "kernel %.4f (min %.4f med %.4f)"%(r["kernel_ms"], r["kernel_ms_min"], r["kernel_ms_median"]), "frac %.3f"%r["frac"], "verified", d.get("verified"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
}
B="--no-cpu-baseline --no-fp32 --no-chain --no-series"
for rep in 1 2; do
  for v in base nt1 nt2 nt3; do
    L=""; [ $v != base ] && L=$PWD/tools/exp/libexp_$v.so
    MFM_LIB=$L timeout 300 python bench.py $B --steps 40 --warmup 5 --config cfg3_1024ch --channels-per-gpu 1024 > $O/c1024_${v}_$rep.json 2> $O/c1024_${v}_$rep.err; summ $O/c1024_${v}_$rep.json
  done
done
for rep in 1 2; do
  for v in base nt1 nt3; do
    L=""; [ $v != base ] && L=$PWD/tools/exp/libexp_$v.so
    MFM_LIB=$L timeout 300 python bench.py $B --steps 200 --warmup 10 > $O/c64_${v}_$rep.json 2> $O/c64_${v}_$rep.err; summ $O/c64_${v}_$rep.json
  done
done
for v in base nt1 nt3; do
  L=""; [ $v != base ] && L=$PWD/tools/exp/libexp_$v.so
  export MFM_LIB=$L
  P="python3 bench.py --config cfg3_1024ch --channels-per-gpu 1024 --steps 6 --warmup 2 --settle-seconds 0.3 $B"
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_${v}_fetch -o f -- $P > $O/pmc_${v}_fetch.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_${v}_write -o w -- $P > $O/pmc_${v}_write.log 2>&1
done
unset MFM_LIB
python3 - <<'PY'
import csv,glob
for v in ("base","nt1","nt3"):
    r={}
    try:
        for name,d in (("FETCH_SIZE","fetch"),("WRITE_SIZE","write")):
            fs=glob.glob(f"gpurun_out/r04f/pmc_{v}_{d}/**/*counter_collection.csv",recursive=True)
            x=[float(x["Counter_Value"]) for x in csv.DictReader(open(fs[0])) if "channel_kernel" in x["Kernel_Name"] and x["Counter_Name"]==name]
            x=x[len(x)//2:]; r[name]=sum(x)/len(x)
        print(v, "FETCH_SIZE %.0f KB WRITE_SIZE %.0f KB -> %.1f MB per launch (FETCH x2 + WRITE)" % (r["FETCH_SIZE"], r["WRITE_SIZE"], (2*r["FETCH_SIZE"]+r["WRITE_SIZE"])*1024/1e6))
    except Exception as e:
        print(v, "ERR", e)
PY
rm -rf $O/pmc_*_fetch $O/pmc_*_write
timeout 300 python3 - <<'PY'
import json, sys, os
sys.path.insert(0, os.getcwd())
import bench
from __graft_entry__ import load_package
pkg = load_package()
fs, decim, taps, offs, gains = pkg.synth.plan("cfg2_64ch", nr_channels=64)
print(json.dumps(bench.end_to_end(pkg, fs, decim, taps, offs, gains), indent=1))
PY
