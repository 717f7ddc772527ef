#!/bin/bash
# One rocprofv3 counter pass over the headline bench (1 step, no warm-up, no CPU baseline): tools/pmc_pass.sh NAME "COUNTERS" [bench args]
# Counters are collected in a run of their own (never together with tracing); the per-kernel sums land in gpurun_out/pmc_NAME.csv
name=$1; shift; ctrs=$1; shift
repo=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_$name
rocprofv3 --pmc $ctrs --output-format csv -d /tmp/pmc_$name -o $name -- python $repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > /tmp/pmc_$name.log 2>&1
cd $repo
python - "$name" <<'PY'
import csv, glob, sys, collections
name = sys.argv[1]
files = glob.glob(f"/tmp/pmc_{name}/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in files:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:70]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        cnt[(k, r["Counter_Name"])] += 1
with open(f"gpurun_out/pmc_{name}.csv", "w") as out:
    out.write("kernel,counter,sum_over_dispatches,dispatches\n")
    for k in sorted(agg):
        for c in sorted(agg[k]):
            out.write(f"\"{k}\",{c},{agg[k][c]:.0f},{cnt[(k, c)]}\n")
print(open(f"gpurun_out/pmc_{name}.csv").read()[:3000])
PY
