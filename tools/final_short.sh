#!/bin/bash
# The short form of final_profiles.sh (when the round's GPU minutes are nearly spent): tools/final_short.sh TAG -> gpurun_out/TAG_*
# the GPU suite, the headline line, its kernel statistics and launch rows, the two traffic passes, one line each of C1 / C2 / C5
tag=$1
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -5 > gpurun_out/${tag}_gputests.log; cat gpurun_out/${tag}_gputests.log
python bench.py > gpurun_out/${tag}_bench_c3_64gib.json 2> gpurun_out/${tag}_bench_c3.err < /dev/null
tail -c 400 gpurun_out/${tag}_bench_c3_64gib.json; echo
timeout 600 tools/kernel_stats.sh ${tag}_c3 --workload c3 --steps 3 --warmup 2 > /dev/null 2>&1 < /dev/null
timeout 600 tools/pmc_pass.sh ${tag}_fetch "FETCH_SIZE" > /dev/null 2>&1 < /dev/null
timeout 600 tools/pmc_pass.sh ${tag}_write "WRITE_SIZE" > /dev/null 2>&1 < /dev/null
sleep 15
python bench.py --workload c1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null < /dev/null | tail -1 > gpurun_out/${tag}_bench_c1.json
python bench.py --workload c2 --no-cpu-baseline 2>/dev/null < /dev/null | tail -1 > gpurun_out/${tag}_bench_c2.json
sleep 15
python bench.py --workload c5 --warmup 2 --no-cpu-baseline 2>/dev/null < /dev/null | tail -1 > gpurun_out/${tag}_bench_c5.json
ls -la gpurun_out/ | grep ${tag}
