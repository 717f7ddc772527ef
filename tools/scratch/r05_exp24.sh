#!/bin/bash
# -r in the wave kernels: GPU tests, Russian text rates, fuzz
mkdir -p gpurun_out/r05e
export PYTHONPATH=/root/repo:/root/repo/tests
timeout 900 python -m pytest tests/test_gpu_wave.py -x -q -m gpu > gpurun_out/r05e/wave_tests.log 2>&1; echo "wave tests rc $?" 
tail -5 gpurun_out/r05e/wave_tests.log
timeout 600 python tools/gpu_text.py 256 > gpurun_out/r05e/text.log 2>&1; echo "text rc $?"
cat gpurun_out/r05e/text.log
timeout 400 python tools/gpu_fuzz.py 300 501 > gpurun_out/r05e/fuzz.log 2>&1; echo "fuzz rc $?"
tail -3 gpurun_out/r05e/fuzz.log
