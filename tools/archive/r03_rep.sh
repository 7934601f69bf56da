cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r03rep
for i in 1 2 3 4 5 6; do
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-fp32 --no-chain 2>/dev/null | tail -1 > gpurun_out/r03rep/run$i.json
python3 - <<PY
import json
d=json.load(open("gpurun_out/r03rep/run$i.json"))
r=d["roofline"]
print($i, round(d["ms_per_step"],4), round(r["frac"],4), r["kernel_ms"], r["kernel_ms_min"], r["kernel_ms_median"], r["kernel_ms_p95"], r.get("timed_launches"), d["verified"])
PY
done
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r03rep/full.json
python3 -c "
import json
d=json.load(open('gpurun_out/r03rep/full.json')); r=d['roofline']
print('full', d['ms_per_step'], r['frac'], r['kernel_ms'], r['kernel_ms_min'], r['kernel_ms_median'], r['kernel_ms_p95'])"
