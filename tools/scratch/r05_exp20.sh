cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_fast_replay.py tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py -q -m gpu -x 2>&1 | tail -5 > gpurun_out/e20_tests.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/e20_bench.json 2> gpurun_out/e20_bench.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/e20_bench2.json 2> gpurun_out/e20_bench2.err
timeout 300 python bench.py --workload c2 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/e20_bench_c2.json
timeout 400 python tools/gpu_fuzz.py 300 503 > gpurun_out/e20_fuzz.log 2>&1
