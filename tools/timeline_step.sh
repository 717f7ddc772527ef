#!/bin/bash
# Host-side marks of ONE timed step (SX_TIMELINE=1: milliseconds since the scan call began): tools/timeline_step.sh TAG [bench args]
tag=${1:-tl}; shift
SX_TIMELINE=1 python bench.py --no-cpu-baseline --no-alone --steps 1 --warmup 2 "$@" > gpurun_out/${tag}_tl.json 2> gpurun_out/${tag}_tl.err
python - "$tag" <<'PY'
import sys
tag = sys.argv[1]
lines = [l.rstrip() for l in open(f'gpurun_out/{tag}_tl.err') if l.startswith('[tl')]
idx = [i for i, l in enumerate(lines) if 'scan_common:' in l]
s = idx[2] if len(idx) > 2 else idx[-1]
e = idx[3] if len(idx) > 3 else len(lines)
open(f'gpurun_out/{tag}_timeline.txt', 'w').write("\n".join(lines[s:e]) + "\n")
print("\n".join(lines[s:e][:150]))
PY
