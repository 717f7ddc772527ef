cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_fast_replay.py -q -m gpu -x 2>&1 | tail -15 > gpurun_out/e11_tests.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py -q -m gpu -x 2>&1 | tail -4 >> gpurun_out/e11_tests.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/e11_bench.json 2> gpurun_out/e11_bench.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/e11_bench2.json 2> gpurun_out/e11_bench2.err
SX_FAST_REPLAY=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/e11_bench_nofast.json 2> gpurun_out/e11_bench_nofast.err
