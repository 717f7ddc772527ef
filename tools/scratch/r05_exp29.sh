#!/bin/bash
mkdir -p gpurun_out/r05i
export PYTHONPATH=/root/repo:/root/repo/tests
timeout 120 python tools/scratch/r05_same_dbg6.py 2>&1 | grep -v 'amdgpu.ids\|^positions\|^wrong' | cut -c1-300; echo "dbg rc $?"
timeout 300 python tools/gpu_text.py 256 russian > gpurun_out/r05i/text.log 2>&1; cat gpurun_out/r05i/text.log | grep -v amdgpu
for b in 2 8; do echo "SX_WAVE_BATCHES=$b"; SX_WAVE_BATCHES=$b timeout 120 python tools/gpu_text.py 256 russian 2>&1 | grep -v amdgpu | grep -e '-r'; done
SX_FUZZ_TRACE=gpurun_out/r05i/fuzz_trace.txt timeout 460 python tools/gpu_fuzz.py 400 503 > gpurun_out/r05i/fuzz.log 2>&1; echo "fuzz rc $?"; tail -2 gpurun_out/r05i/fuzz.log | cut -c1-600; cat gpurun_out/r05i/fuzz_trace.txt | cut -c1-600
