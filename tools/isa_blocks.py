#!/usr/bin/env python3
"""Basic blocks of one kernel of an ISA listing (hipcc -S --cuda-device-only), with instruction counts per block and where each block
branches: tools/isa_blocks.py FILE.s KERNEL_SUBSTRING [min_valu]  — to read a hot loop's cost (VALU, SGPR-spill traffic) off the listing."""
import re, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2]
minv = int(sys.argv[3]) if len(sys.argv) > 3 else 0
for m in re.finditer(r"\n(_Z\w+): +; @", txt):
    if flt not in m.group(1):
        continue
    body = txt[m.end():txt.find(".Lfunc_end", m.end())]
    blocks, cur = [], ["entry", []]
    for line in body.split("\n"):
        lm = re.match(r"^(\.LBB\d+_\d+):", line)
        if lm:
            blocks.append(cur); cur = [lm.group(1), []]
        elif re.match(r"^\s+[a-z]", line):
            cur[1].append(line.strip())
    blocks.append(cur)
    print(m.group(1)[-60:], len(blocks), "blocks")
    for name, ins in blocks:
        v = sum(1 for i in ins if i.startswith("v_"))
        if v < minv:
            continue
        s = sum(1 for i in ins if i.startswith("s_"))
        rl = sum(1 for i in ins if i.startswith("v_readlane") or i.startswith("v_writelane"))
        ld = sum(1 for i in ins if i.startswith("buffer_load") or i.startswith("global_load"))
        st = sum(1 for i in ins if "store" in i or "atomic" in i)
        br = [i.split()[-1] for i in ins if i.startswith("s_cbranch") or i.startswith("s_branch")]
        print(f"{name:12s} valu {v:4d} (lane-spill {rl:3d}) salu {s:4d} loads {ld:2d} stores {st:2d} -> {' '.join(br)}")
    break
