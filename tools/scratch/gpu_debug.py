#!/usr/bin/env python3
"""Debug aid: for one failing end-to-end case, print the first differing output lines and
any device-run vs oracle-run mismatch per mission and file."""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import refconfig as rc, stringsext_amd as sx, sxo_binding as sxo
from product_harness import run_cli_product
from test_host_logic import soup, synth

opts = {'chars_min': '5', 'output_line_len': '10'}
encs = ['utf-8,3,All,All', 'utf-16le,,,Asian', 'ascii,5']
seed = 1022710673
rng = random.Random(seed)
ms = rc.missions(encodings=encs, **opts)
files = [synth(rng, 150_000, 1 / 300), soup(rng, 20_001), synth(rng, 4096 * 3 + 1, 1 / 100), b"", synth(rng, 50_000, 1 / 2000)]
want = sxo.run_cli(ms, files, radix="x").split(b"\n")
got = run_cli_product(ms, files, radix="x", device=0).split(b"\n")
print("lines", len(want), len(got))
for i, (a, b) in enumerate(zip(want, got)):
    if a != b:
        print("first diff at line", i); print(" want", want[i - 1:i + 3]); print(" got ", got[i - 1:i + 3]); break
stream = 0
for fi, f in enumerate(files):
    for mi, m in enumerate(ms):
        m1 = dict(m, mission_id=0)
        mc = max(1, min(m["chars_min_nb"], m["output_line_char_nb_max"]))
        for generic in (False, True):
            sc = sx.Scanner([m1], device=0, generic_kernels=generic)
            if f:
                d = sc.alloc(len(f)); sc.upload(d, f)
                dev = sc.device_runs(0, d, len(f), stream & 1, mc)
                sc.free(d)
            else:
                dev = []
            sc.close()
            ora = sxo.runs(m1, f, stream_parity=stream & 1, min_chars=mc)
            if dev != ora:
                sd, so = set(dev), set(ora)
                print(f"file {fi} mission {mi} generic={generic}: device {len(dev)} oracle {len(ora)}; only device {sorted(sd - so)[:5]} only oracle {sorted(so - sd)[:5]}")
                for (a, b, c) in sorted(so ^ sd)[:3]:
                    print("   bytes", a, f[max(0, a - 4):b + 4].hex())
    stream += len(f)
print("done")
