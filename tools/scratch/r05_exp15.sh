cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_wave.py -q -m gpu -x 2>&1 | tail -25 > gpurun_out/e15_tests.log
