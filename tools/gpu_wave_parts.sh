#!/bin/bash
# Where the wave count pass spends its time: the product next to builds that stop after the classification (1), after the windows'
# masks are out of LDS (2), and that run the state machine without descriptors (3) — their results are garbage, only the count
# kernel's duration in a kernel trace is read.  tools/gpu_wave_parts.sh "ENC[,,,UBF]" [GiB] [n]
enc=${1:-ascii}; gib=${2:-4}; export EXP_N=${3:-10}
repo=$(pwd); export TMPDIR=/tmp
line="$enc n=$EXP_N ${gib}GiB, us per launch:"
for v in 0 1 2 3; do
  lib=""; [ $v != 0 ] && lib=$repo/build_exp/wvexp$v/libstringsext_amd.so
  [ $v != 0 ] && [ ! -f "$lib" ] && continue
  rm -rf /tmp/wvp_$v
  (cd /tmp && SX_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/wvp_$v -o p -- python $repo/tools/gpu_wave_exp.py "$enc" $gib > /tmp/wvp_$v.log 2>&1)
  f=$(find /tmp/wvp_$v -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && line="$line v$v $(python3 - "$f" <<'PY'
import csv, sys
out = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "wave_replay_kernel<0" in n: out.append("count %.0f" % (float(r["AverageNs"]) / 1e3))
    elif "wave_emit" in n: out.append("emit %.0f" % (float(r["AverageNs"]) / 1e3))
print(" ".join(out))
PY
) |"
done
echo "$line"
