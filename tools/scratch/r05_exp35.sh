#!/bin/bash
mkdir -p gpurun_out/r05r
export PYTHONPATH=/root/repo:/root/repo/tests
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r05r/gputests.log 2>&1; echo "gpu tests rc $?"; tail -3 gpurun_out/r05r/gputests.log
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r05r/bench_c3_64gib.json; python -c "
import json; j=json.loads(open('gpurun_out/r05r/bench_c3_64gib.json').read()); r=j['roofline']; print(j['value'], j['ms_per_step'], r['frac'], r['frac_of_step_wall'], r['traffic'])"
