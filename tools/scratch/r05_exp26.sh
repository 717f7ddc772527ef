#!/bin/bash
export PYTHONPATH=/root/repo:/root/repo/tests
for b in 1 2 4; do echo "SX_WAVE_BATCHES=$b"; SX_WAVE_BATCHES=$b timeout 300 python tools/gpu_text.py 256 russian 2>&1 | grep -v amdgpu; done
echo "SX_WAVE_DESC=0 batches 1"; SX_WAVE_DESC=0 SX_WAVE_BATCHES=1 timeout 300 python tools/gpu_text.py 256 russian 2>&1 | grep -v amdgpu
