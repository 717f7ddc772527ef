#!/bin/bash
mkdir -p gpurun_out/r05j
export PYTHONPATH=/root/repo:/root/repo/tests
echo "--- -r on the lane-per-region path (SX_WAVE_SAME=0), Russian text"; SX_WAVE_SAME=0 timeout 600 python tools/gpu_text.py 256 russian 2>&1 | grep -v amdgpu | tee gpurun_out/r05j/text_russian_same0.log
echo "--- -q 255"; timeout 600 python tools/gpu_text.py 256 q255 2>&1 | grep -v amdgpu | tee gpurun_out/r05j/text_q255.log
tools/sort_reach.sh r05j > /dev/null 2>&1; cat gpurun_out/r05j_sort_reach.txt | cut -c1-330
