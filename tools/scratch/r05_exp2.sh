cd $GRAFT_REPO_ROOT
timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/e2_bench.json 2> gpurun_out/e2_bench.err
SX_TIMING=1 SX_TIMING2=1 timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/e2_bench_t.json 2> gpurun_out/e2_bench_t.err
