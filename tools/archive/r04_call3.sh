#!/bin/bash
# round 4, third GPU call: v_rcp_f32's table + the division proof on it; the coalescing / overlap tests; the default bench line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c; rm -rf $O; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/rcp_check tools/rcp_check.hip && timeout 300 /tmp/rcp_check $O/rcp_dev.bin > $O/rcp_check.txt 2>&1; echo "rcp_check rc=$?"; cat $O/rcp_check.txt
gcc -O2 -ffp-contract=off -march=native -o /tmp/div_proof tools/div_proof.c -lm -lpthread
for st in 1 2; do timeout 300 /tmp/div_proof $st 32 - $O/rcp_dev.bin > $O/div_proof_steps$st.txt 2>&1; echo "rc=$?" >> $O/div_proof_steps$st.txt; tail -4 $O/div_proof_steps$st.txt; done
rm -f $O/rcp_dev.bin
timeout 900 python -m pytest tests/test_coalesce.py tests/test_group.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"; tail -2 $O/bench_n1.err
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r04c/bench_n1.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("value %.4g ms/step %.4f kernel %.4f frac %.3f verified %s" % (d["value"], d["ms_per_step"], r["kernel_ms"], r["frac"], d["verified"]))
for row in d["block_series"]["series"]:
    for m in row:
        if isinstance(row[m], dict):
            print(row["block_samples"], m, {k: (round(v, 3) if isinstance(v, float) else v) for k, v in row[m].items()})
print(json.dumps(d["end_to_end"], indent=1))
print(json.dumps(d["cpu_baseline"], indent=1))
PY
