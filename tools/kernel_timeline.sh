#!/bin/bash
# rocprofv3 --kernel-trace over one bench run; prints the last step's kernels as a timeline (start offset, duration, name)
# usage: tools/kernel_timeline.sh NAME [bench args]   -> gpurun_out/NAME_timeline.txt (everything from the first scan launch on)
name=$1; shift
repo=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_$name
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$name -o $name -- python $repo/bench.py --no-cpu-baseline "$@" > /tmp/kt_$name.log 2>/tmp/kt_$name.err < /dev/null
cd $repo
f=$(find /tmp/kt_$name -name '*kernel_trace.csv' | head -1)
if [ -z "$f" ]; then echo "no kernel_trace.csv"; tail -5 /tmp/kt_$name.err; exit 1; fi
python - "$f" "$name" "${SX_TL_MISSIONS:-1}" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last step: from the last scan_kernel launch on
# the last timed step: bench.py runs the scan kernels once more, alone, after the timed region -> the group before the last
scans = [i for i, r in enumerate(rows) if "scan_kernel" in r["Kernel_Name"]]
nm = int(sys.argv[3]) if len(sys.argv) > 3 else 1            # scan launches per step (= missions)
end = len(rows)
idx = scans[0]
t0 = int(rows[idx]["Start_Timestamp"])
with open(f"gpurun_out/{sys.argv[2]}_timeline.txt", "w") as out:
    for r in rows[idx:end]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        nm = r["Kernel_Name"].split("(")[0].replace("void ", "")[-60:]
        out.write(f"{(s - t0) / 1e6:9.3f} ms  +{(e - s) / 1e6:8.3f} ms  q{r.get('Queue_Id', '?')}  {nm}\n")
print("scan launches at lines", [i - idx + 1 for i in scans])
PY
