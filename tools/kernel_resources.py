#!/usr/bin/env python3
"""Registers, spills, scratch, LDS and occupancy of every kernel of one .hip file of stringsext_amd/csrc (no GPU needed):
tools/kernel_resources.py sx_wave_dev.hip [extra hipcc flags]"""
import os, re, subprocess, sys
csrc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "stringsext_amd", "csrc")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", sys.argv[1], "-o", "/dev/null",
       "-Rpass-analysis=kernel-resource-usage"] + sys.argv[2:]
err = subprocess.run(cmd, cwd=csrc, capture_output=True, text=True).stderr
cur, rows = None, {}
for line in err.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1); rows[cur] = {}; continue
    m = re.search(r"remark: +([A-Za-z ]+?(?: \[[^\]]*\])?): (\S+) \[-Rpass", line)
    if cur and m:
        rows[cur][m.group(1).strip()] = m.group(2)
if not rows:
    print(err[-3000:])
for k, v in rows.items():
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(.*", "", name)
    if "rocprim" in name:
        continue
    g = lambda key: v.get(key, "?")
    print(f"{name[:64]:64s} VGPR {g('VGPRs'):>4} AGPR {g('AGPRs'):>3} SGPR {g('TotalSGPRs'):>4} spillS {g('SGPRs Spill'):>3} spillV {g('VGPRs Spill'):>3} "
          f"scratch {g('ScratchSize [bytes/lane]'):>5} occ {g('Occupancy [waves/SIMD]'):>2} LDS {g('LDS Size [bytes/block]'):>6}")
