#!/bin/bash
python -m pytest tests/test_gpu_wave.py tests/test_gpu_dbcs.py -x -q > gpurun_out/s2b_tests.log 2>&1 < /dev/null; tail -3 gpurun_out/s2b_tests.log
for e in big5,,,Cjk shift_jis,,,Cjk euc-kr,,,Cjk; do python tools/gpu_wave_exp.py $e 4 2>&1 | tail -1; done > gpurun_out/s2b_wave_exp.log
cat gpurun_out/s2b_wave_exp.log
