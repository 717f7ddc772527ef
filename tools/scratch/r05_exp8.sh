cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 > gpurun_out/e8_tests.log
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/e8_bench.json 2> gpurun_out/e8_bench.err
