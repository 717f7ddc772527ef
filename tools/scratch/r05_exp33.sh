#!/bin/bash
mkdir -p gpurun_out/r05l
cd /tmp && export TMPDIR=/tmp PYTHONPATH=/root/repo:/root/repo/tests
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_same -o same -- python /root/repo/tools/gpu_text.py 256 russian > /root/repo/gpurun_out/r05l/prof_run.log 2>&1
find /tmp/prof_same -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} /root/repo/gpurun_out/r05l/russian_kernel_stats.csv
head -9 /root/repo/gpurun_out/r05l/russian_kernel_stats.csv | cut -c1-200
