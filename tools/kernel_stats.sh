#!/bin/bash
# rocprofv3 --kernel-trace --stats over one bench run: tools/kernel_stats.sh NAME [bench args]
# -> gpurun_out/NAME_kernel_stats.csv (per-kernel totals and averages), NAME_launch_rows.csv (per kernel and class of launch:
# warm-up, timed, alone) and NAME_bench_under_rocprof.json
name=$1; shift
repo=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks_$name
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$name -o $name -- python $repo/bench.py --no-cpu-baseline "$@" > /tmp/ks_$name.log 2>/tmp/ks_$name.err < /dev/null
cd $repo
f=$(find /tmp/ks_$name -name '*kernel_stats.csv' | head -1)
if [ -z "$f" ]; then echo "no kernel_stats.csv under /tmp/ks_$name"; tail -5 /tmp/ks_$name.err; exit 1; fi
cp "$f" gpurun_out/${name}_kernel_stats.csv
tail -1 /tmp/ks_$name.log > gpurun_out/${name}_bench_under_rocprof.json
# the same trace launch by launch: warm-up / timed steps / alone (bench.py's marker launches split it) -> the roofline fraction to the digit
t=$(find /tmp/ks_$name -name '*kernel_trace.csv' | head -1)
if [ -n "$t" ]; then python tools/launch_rows.py "$t" gpurun_out/${name}_bench_under_rocprof.json > gpurun_out/${name}_launch_rows.csv 2>/tmp/ks_$name.rows.err || tail -3 /tmp/ks_$name.rows.err; fi
head -16 gpurun_out/${name}_kernel_stats.csv | cut -c1-220
