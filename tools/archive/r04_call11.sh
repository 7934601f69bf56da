#!/bin/bash
# round 4: which LDS access conflicts.  Three builds of the second-generation kernel - shipped (T[256] | dT[256],
# two 4-byte reads per output), lut1 ({T, dT} pairs, one 8-byte read), lut2 (counter build: table reads at one address per bank,
# wrong PCM) - each timed and each run under the SQ LDS counters.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04lds; rm -rf $O; mkdir -p $O
summ() { python3 - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print(sys.argv[1].split('/')[-1], "%.4g"%d["value"], "ms/step %.4f"%d["ms_per_step"], "kernel %.4f (min %.4f med %.4f)"%(r["kernel_ms"], r["kernel_ms_min"], r["kernel_ms_median"]), "frac %.3f"%r["frac"], "verified", d.get("verified"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
}
VARIANTS=${VARIANTS:-"base lut1 lut2"}
B="--no-cpu-baseline --no-fp32 --no-chain --no-series"
for rep in 1 2 3; do
  for v in $VARIANTS; do
    L=""; [ $v != base ] && L=$PWD/tools/exp/libexp_$v.so
    MFM_LIB=$L timeout 300 python bench.py $B --steps 200 --warmup 10 > $O/c64_${v}_$rep.json 2> $O/c64_${v}_$rep.err; summ $O/c64_${v}_$rep.json
  done
done
[ -z "$SKIP_TESTS" ] && MFM_LIB=$PWD/tools/exp/libexp_lut1.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_coalesce.py -m gpu -x -q 2>&1 | tail -2
P="python3 bench.py --steps 8 --warmup 3 --settle-seconds 0.3 $B"
for v in $VARIANTS; do
  L=""; [ $v != base ] && L=$PWD/tools/exp/libexp_$v.so
  export MFM_LIB=$L
  timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/pmc_$v -o p -- $P > $O/pmc_$v.log 2>&1
  python3 - $O/pmc_$v $v <<'PY'
import csv, glob, sys, collections
d, v = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mfm_channel_kernel_v3" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: sum(x[len(x)//2:]) / max(1, len(x[len(x)//2:])) for k, x in acc.items()}
print(v, {k: "%.4g" % a for k, a in sorted(out.items())})
if "SQ_LDS_IDX_ACTIVE" in out:
    print(v, "conflict / active = %.3f" % (out["SQ_LDS_BANK_CONFLICT"] / out["SQ_LDS_IDX_ACTIVE"]))
PY
done
unset MFM_LIB
