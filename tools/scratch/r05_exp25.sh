#!/bin/bash
# where the time of `-e utf-8 -u Cyrillic -r` on Russian text goes
mkdir -p gpurun_out/r05f
export PYTHONPATH=/root/repo:/root/repo/tests
cd /tmp && export TMPDIR=/tmp
SX_TIMING=1 SX_TIMING2=1 timeout 300 python /root/repo/tools/gpu_text.py 256 utf-8 > /root/repo/gpurun_out/r05f/timing.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_same -o same -- python /root/repo/tools/gpu_text.py 256 russian utf-16le > /root/repo/gpurun_out/r05f/prof_run.log 2>&1
find /tmp/prof_same -name '*kernel_stats*' | head -1 | xargs -I{} cp {} /root/repo/gpurun_out/r05f/kernel_stats.csv
head -12 /root/repo/gpurun_out/r05f/kernel_stats.csv | cut -c1-220
grep -c . /root/repo/gpurun_out/r05f/timing.log
grep -v "^\[sx\] replay" /root/repo/gpurun_out/r05f/timing.log | tail -45 | cut -c1-250
