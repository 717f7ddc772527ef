#!/bin/bash
# Does any BASELINE configuration, or an alias filter on the headline image, still reach the radix sort of the scan kernels' general path
# (sx_kernels.hip: records in atomic order -> rocprim radix sort)?  One bench run per case under rocprofv3 --kernel-trace, the sort's kernels
# counted by name.  tools/sort_reach.sh TAG -> gpurun_out/TAG_sort_reach.txt
tag=$1
repo=$(pwd)
out=$repo/gpurun_out/${tag}_sort_reach.txt
: > $out
cd /tmp && export TMPDIR=/tmp
run() {   # NAME bench-args...
  name=$1; shift
  rm -rf /tmp/sr_$name
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sr_$name -o $name -- python $repo/bench.py --no-cpu-baseline --steps 2 --warmup 1 "$@" > /tmp/sr_$name.log 2>/dev/null < /dev/null
  f=$(find /tmp/sr_$name -name '*kernel_stats.csv' | head -1)
  if [ -z "$f" ]; then echo "$name: no kernel stats" >> $out; return; fi
  python3 - "$f" "$name" "$*" /tmp/sr_$name.log >> $out <<'PY'
import csv, json, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if any(k in r["Name"].lower() for k in ("radix", "onesweep", "sort_"))]
try: ms = json.loads(open(sys.argv[4]).read().strip().splitlines()[-1])["ms_per_step"]
except Exception: ms = None
print(f"{sys.argv[2]} ({sys.argv[3]}): sort kernels {len(rows)}, launches {sum(int(r['Calls']) for r in rows)}, total {sum(int(r['TotalDurationNs']) for r in rows) / 1e6:.3f} ms; bench ms_per_step {ms}")
PY
}
run c1 --workload c1
run c1_steps6 --workload c1 --steps 6
run c2 --workload c2
run c3
run c5 --workload c5
run c5_steps4 --workload c5 --steps 4
for f in Cjk Hangul Kana All Latin Asian; do run c3_ubf_$f --ubf $f --gib 16; done
run c3_ubf_Asian_steps4 --ubf Asian --gib 16 --steps 4
run c3_generic --generic-kernels --gib 16
cat $out
