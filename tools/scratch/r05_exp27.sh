#!/bin/bash
mkdir -p gpurun_out/r05g
export PYTHONPATH=/root/repo:/root/repo/tests
timeout 900 python -m pytest tests/test_gpu_wave.py -x -q -m gpu -k "same_unicode" > gpurun_out/r05g/wave_tests.log 2>&1; echo "wave tests rc $?"; tail -3 gpurun_out/r05g/wave_tests.log
timeout 600 python tools/gpu_text.py 256 russian > gpurun_out/r05g/text.log 2>&1; cat gpurun_out/r05g/text.log | grep -v amdgpu
timeout 300 python tools/gpu_fuzz.py 200 502 > gpurun_out/r05g/fuzz.log 2>&1; echo "fuzz rc $?"; tail -2 gpurun_out/r05g/fuzz.log
