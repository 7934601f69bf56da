cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r01b
timeout 300 python bench.py > gpurun_out/r01b/bench.json 2> gpurun_out/r01b/bench.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01b/kstats -o k -- python3 bench.py --steps 300 --warmup 150 --no-cpu-baseline --no-fp32 > gpurun_out/r01b/kstats.log 2>&1
timeout 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r01b/fetch -o f -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-fp32 > gpurun_out/r01b/fetch.log 2>&1
timeout 240 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r01b/write -o w -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-fp32 > gpurun_out/r01b/write.log 2>&1
./tools/ubench_overlap > gpurun_out/r01b/ubench_overlap.txt 2>&1
ls -R gpurun_out/r01b | head -30
# the float path's kernel: HBM traffic of one 2^24-sample block (same two PMC passes)
timeout 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r01b/f32fetch -o f -- python3 tools/bench_f32.py --iters 3 > gpurun_out/r01b/f32fetch.log 2>&1
timeout 240 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r01b/f32write -o w -- python3 tools/bench_f32.py --iters 3 > gpurun_out/r01b/f32write.log 2>&1
