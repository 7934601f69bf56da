#!/bin/bash
# round 4, seventh GPU call: PCM stores with the non-temporal hint (nt2) and 4-byte rotator entries on top (rot4): parity of
# the rot4 build, time and L2-miss traffic at 1024 and 64 channels
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04g; rm -rf $O; mkdir -p $O
MFM_LIB=$PWD/tools/exp/libexp_rot4.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_coalesce.py -m gpu -x -q -k "not discriminator_division" > $O/pytest_rot4.log 2>&1; echo "pytest rot4 rc=$?"; tail -3 $O/pytest_rot4.log
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
for rep in 1 2 3; do
  for v in base nt2 rot4; do
    L=""; [ $v != base ] && L=$PWD/tools/exp/libexp_$v.so
    MFM_LIB=$L timeout 300 python bench.py $B --steps 40 --warmup 5 --config cfg3_1024ch --channels-per-gpu 1024 > $O/c1024_${v}_$rep.json 2> $O/c1024_${v}_$rep.err; summ $O/c1024_${v}_$rep.json
  done
done
for rep in 1 2 3; do
  for v in base nt2 rot4; do
    L=""; [ $v != base ] && L=$PWD/tools/exp/libexp_$v.so
    MFM_LIB=$L timeout 300 python bench.py $B --steps 200 --warmup 10 > $O/c64_${v}_$rep.json 2> $O/c64_${v}_$rep.err; summ $O/c64_${v}_$rep.json
  done
done
for cfgname in c1024 c64; do
for v in base nt2 rot4; do
  L=""; [ $v != base ] && L=$PWD/tools/exp/libexp_$v.so
  export MFM_LIB=$L
  if [ $cfgname = c1024 ]; then P="python3 bench.py --config cfg3_1024ch --channels-per-gpu 1024 --steps 6 --warmup 2 --settle-seconds 0.3 $B"; else P="python3 bench.py --steps 20 --warmup 5 --settle-seconds 0.3 $B"; fi
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_${cfgname}_${v}_fetch -o f -- $P > $O/pmc_${cfgname}_${v}_fetch.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_${cfgname}_${v}_write -o w -- $P > $O/pmc_${cfgname}_${v}_write.log 2>&1
done
done
unset MFM_LIB
python3 - <<'PY' | tee gpurun_out/r04g/traffic.txt
import csv,glob
for c in ("c1024","c64"):
  for v in ("base","nt2","rot4"):
    r={}
    try:
        for name,d in (("FETCH_SIZE","fetch"),("WRITE_SIZE","write")):
            fs=glob.glob(f"gpurun_out/r04g/pmc_{c}_{v}_{d}/**/*counter_collection.csv",recursive=True)
            x=[float(x["Counter_Value"]) for x in csv.DictReader(open(fs[0])) if "channel_kernel" in x["Kernel_Name"] and x["Counter_Name"]==name]
            x=x[len(x)//2:]; r[name]=sum(x)/len(x)
        print(c, v, "FETCH_SIZE %.0f KB WRITE_SIZE %.0f KB -> %.1f MB per launch (FETCH x2 + WRITE)" % (r["FETCH_SIZE"], r["WRITE_SIZE"], (2*r["FETCH_SIZE"]+r["WRITE_SIZE"])*1024/1e6))
    except Exception as e:
        print(c, v, "ERR", e)
PY
rm -rf $O/pmc_*_fetch $O/pmc_*_write
