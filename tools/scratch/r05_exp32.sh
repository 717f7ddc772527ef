#!/bin/bash
mkdir -p gpurun_out/r05k
export PYTHONPATH=/root/repo:/root/repo/tests
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r05k/gpu_tests.log 2>&1; echo "gpu tests rc $?"; tail -4 gpurun_out/r05k/gpu_tests.log
for w in c1 c5 c2; do python bench.py --workload $w --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r05k/bench_$w.json; cut -c1-180 gpurun_out/r05k/bench_$w.json; done
python bench.py 2>/dev/null | tail -1 > gpurun_out/r05k/bench_c3.json; cut -c1-400 gpurun_out/r05k/bench_c3.json
timeout 300 python tools/gpu_text.py 256 russian 2>&1 | grep -e ' -r'
