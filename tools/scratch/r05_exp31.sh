#!/bin/bash
export PYTHONPATH=/root/repo:/root/repo/tests
echo "--- c5 default"; python bench.py --workload c5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200
echo "--- c5 OCC4S=3"; SX_LIB=build_exp/v_occ4s3/libstringsext_amd.so python bench.py --workload c5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200
echo "--- russian default"; timeout 200 python tools/gpu_text.py 256 russian 2>&1 | grep -e ' -r'
echo "--- russian OCCS=3"; SX_LIB=build_exp/v_occs3/libstringsext_amd.so timeout 200 python tools/gpu_text.py 256 russian 2>&1 | grep -e ' -r'
echo "--- c1, c2"; python bench.py --workload c1 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200; python bench.py --workload c2 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200; python bench.py --workload c2 --gib 8 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200
