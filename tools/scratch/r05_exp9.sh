cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py -q -m gpu -x 2>&1 | tail -4 > gpurun_out/e9_tests.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/e9_bench.json 2> gpurun_out/e9_bench.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/e9_bench2.json 2> gpurun_out/e9_bench2.err
