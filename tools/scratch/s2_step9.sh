#!/bin/bash
timeout 200 python -m pytest tests/test_gpu_wave.py tests/test_gpu_dbcs.py -x -q 2>&1 | tail -2
for e in big5,,,Cjk koi8-r,,,Cyrillic utf-8; do timeout 60 python tools/gpu_wave_exp.py $e 4 2>&1 | tail -1; done
EXP_N=4 timeout 60 python tools/gpu_wave_exp.py ascii 4 2>&1 | tail -1
timeout 60 python tools/gpu_text.py 256 2>&1 | tail -3
