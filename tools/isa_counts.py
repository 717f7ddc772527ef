#!/usr/bin/env python3
"""Static instruction counts per kernel of an ISA listing (hipcc -S --cuda-device-only): tools/isa_counts.py FILE.s [name filter]"""
import re, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
PATS = [("valu", r"\n\s+v_"), ("salu", r"\n\s+s_"), ("ds_r", "ds_read"), ("ds_w", "ds_write"), ("flat_ld", "flat_load"), ("flat_st", "flat_store"),
        ("glob_ld", "global_load"), ("glob_st", "global_store"), ("scratch", "scratch_"), ("bperm", "ds_bpermute")]
for m in re.finditer(r"\n(_Z\w+): +; @", txt):
    nm = m.group(1)
    if flt not in nm:
        continue
    j = txt.find(".Lfunc_end", m.end())
    body = txt[m.end():j]
    print(nm[-40:].ljust(40), "lines", body.count("\n"), " ".join(f"{k} {len(re.findall(p, body))}" for k, p in PATS))
