#!/usr/bin/env python3
"""Puts the figures of one tools/final_profiles.sh session into the texts that quote them: tools/fill_docs.py TAG [NGPU NCPU "FUZZ TEXT"]
replaces @C3@, @C3MS@, @C3WALL@, @C3FRAC@, @C1@, @C2MS@, @C5@, @C5MS@, @C5WALL@, @TEXT@, @XCHG@, @NGPU@, @NCPU@, @FUZZ@ in DESIGN.md and README.md
(one-shot: the placeholders are gone afterwards)."""
import json, os, re, sys
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
tag = sys.argv[1]
J = lambda n: json.load(open(os.path.join(root, "profiles", f"{tag}_{n}.json")))
c3, c1, c2, c5 = J("bench_c3_64gib"), J("bench_c1"), J("bench_c2"), J("bench_c5")
g2 = J("bench_2ranks_1gpu_gloo")
text = open(os.path.join(root, "profiles", f"{tag}_text.txt")).read()
tv = re.findall(r"= ([\d.]+) GiB/s", text)
rep = {"@C3@": f"{c3['value']:.0f}", "@C3MS@": f"{c3['ms_per_step']:.1f}", "@C3WALL@": f"{c3['roofline']['frac_of_step_wall']:.2f}",
       "@C3FRAC@": f"{c3['roofline']['frac']:.2f}", "@C1@": f"{c1['value']:.0f}", "@C2MS@": f"{c2['ms_per_step']:.2f}",
       "@C5@": f"{c5['value']:.0f}", "@C5MS@": f"{c5['ms_per_step']:.0f}", "@C5WALL@": f"{c5['roofline']['frac_of_step_wall']:.2f}",
       "@TEXT@": f"{tv[0]}–{tv[1]}" if len(tv) >= 2 else "?", "@XCHG@": f"{g2['exchange_ms_per_step']:.2f}"}
if len(sys.argv) > 4:
    rep.update({"@NGPU@": sys.argv[2], "@NCPU@": sys.argv[3], "@FUZZ@": sys.argv[4]})
for name in ("DESIGN.md", "README.md"):
    p = os.path.join(root, name)
    s = open(p).read()
    for k, v in rep.items():
        s = s.replace(k, v)
    open(p, "w").write(s)
    print(name, "left:", sorted(set(re.findall(r"@[A-Z0-9]+@", s))))
