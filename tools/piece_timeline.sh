#!/bin/bash
# rocprofv3 --kernel-trace --memory-copy-trace over one bench run; the timed region (between bench.py's two marker launches) as one
# timeline of kernels and copies: tools/piece_timeline.sh NAME [bench args] -> gpurun_out/NAME_timeline.txt
name=$1; shift
repo=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pt_$name
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/pt_$name -o $name -- python $repo/bench.py --no-cpu-baseline "$@" > /tmp/pt_$name.log 2>/tmp/pt_$name.err < /dev/null
cd $repo
tail -1 /tmp/pt_$name.log | cut -c1-200
k=$(find /tmp/pt_$name -name '*kernel_trace.csv' | head -1)
m=$(find /tmp/pt_$name -name '*memory_copy_trace.csv' | head -1)
python - "$k" "$m" "$name" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K q%s " % r.get("Queue_Id", "?") + r["Kernel_Name"].split("(")[0].replace("void ", "")[-70:]) for r in rows]
marks = [e for e in ev if "fill_kernel" in e[2]][-2:]
try:
    for r in csv.DictReader(open(sys.argv[2])):
        n = int(r.get("Bytes", r.get("Size", 0)) or 0)
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C %s %d bytes" % (r.get("Direction", r.get("Name", "?")), n)))
except Exception as e:
    print("no copy trace:", e)
ev.sort()
t0, t1 = marks[0][0], marks[1][1]
with open(f"gpurun_out/{sys.argv[3]}_timeline.txt", "w") as out:
    for s, e, n in ev:
        if t0 <= s <= t1 and (e - s) >= 20000:      # >= 20 us
            out.write(f"{(s - t0) / 1e6:9.3f} ms  +{(e - s) / 1e6:8.3f} ms  {n}\n")
print("timed region", (t1 - t0) / 1e6, "ms")
PY
