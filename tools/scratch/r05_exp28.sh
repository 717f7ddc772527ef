#!/bin/bash
mkdir -p gpurun_out/r05h
export PYTHONPATH=/root/repo:/root/repo/tests
timeout 200 python tools/scratch/r05_same_dbg6.py 2>&1 | grep -v 'amdgpu.ids\|^positions\|^wrong' | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_wave.py -x -q -m gpu > gpurun_out/r05h/wave_tests.log 2>&1; echo "wave tests rc $?"; tail -3 gpurun_out/r05h/wave_tests.log
timeout 600 python tools/gpu_text.py 256 > gpurun_out/r05h/text.log 2>&1; cat gpurun_out/r05h/text.log | grep -v amdgpu
timeout 700 python tools/gpu_fuzz.py 600 503 > gpurun_out/r05h/fuzz.log 2>&1; echo "fuzz rc $?"; tail -2 gpurun_out/r05h/fuzz.log | cut -c1-600
