#!/bin/bash
# round 3, session 2, step 1: GPU suite + the wave kernels alone after the "dull call" skip
python -m pytest tests -m gpu -x -q > gpurun_out/s2a_tests.log 2>&1 < /dev/null; tail -3 gpurun_out/s2a_tests.log
for e in big5,,,Cjk utf-8 koi8-r,,,Cyrillic shift_jis,,,Cjk; do python tools/gpu_wave_exp.py $e 4 2>&1 | tail -1; done > gpurun_out/s2a_wave_exp.log
cat gpurun_out/s2a_wave_exp.log
python tools/gpu_text.py > gpurun_out/s2a_text.txt 2>&1 < /dev/null; tail -3 gpurun_out/s2a_text.txt
python bench.py --workload c1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null < /dev/null | tail -1 > gpurun_out/s2a_c1.json; head -c 300 gpurun_out/s2a_c1.json; echo
SX_WAVE_BYTES_PER_RUN=1000 SX_TIMING=1 python bench.py --workload c5 --no-cpu-baseline 2> gpurun_out/s2a_c5_wave.err < /dev/null | tail -1 > gpurun_out/s2a_c5_wave.json; head -c 300 gpurun_out/s2a_c5_wave.json; echo
