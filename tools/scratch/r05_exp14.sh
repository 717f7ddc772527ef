cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py tests/test_gpu_fast_replay.py -q -m gpu -x 2>&1 | tail -4 > gpurun_out/e14_tests.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/e14_bench.json 2> gpurun_out/e14_bench.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/e14_bench2.json 2> gpurun_out/e14_bench2.err
SX_SMALL_COPY=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/e14_bench_nosmall.json 2> gpurun_out/e14_bench_nosmall.err
SX_TIMELINE=1 timeout 200 python bench.py --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/e14_tl.json 2> gpurun_out/e14_tl.err
