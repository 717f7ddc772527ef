"""Replay one case of tools/gpu_fuzz.py.  usage: tools/gpu_repro.py CASE_SEED [host|dev|case] [KEY=VALUE switches...]
host/dev force where stage B runs, `case` (default) takes the case's own choice; switches given here replace the case's."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fuzz_case
import sxo_binding as sxo
from product_harness import run_cli_product

c = fuzz_case.make(int(sys.argv[1]))
mode = sys.argv[2] if len(sys.argv) > 2 else "case"
sw = dict(a.split("=", 1) for a in sys.argv[3:]) if len(sys.argv) > 3 else c["switches"]
replay = c["replay"] if mode == "case" else mode == "dev"
print(fuzz_case.describe(c), "\n  running with replay =", replay, "switches =", sw)
want = sxo.run_cli(c["missions"], c["files"], radix="x", flush_at_eof=c["flush"])
fuzz_case.set_switches(sw)
got = run_cli_product(c["missions"], c["files"], radix="x", chunk_bytes=c["chunk"], device=0, subchunk_bytes=c["sub"],
                      flush_at_eof=c["flush"], device_replay=replay, generic_kernels=c["generic"])
print("gpu path equal:", got == want)
if got != want:
    gl, wl = got.split(b"\n"), want.split(b"\n")
    print("lines", len(gl), len(wl))
    for i, (a, b) in enumerate(zip(gl, wl)):
        if a != b:
            print("first differing line", i)
            for l in gl[max(0, i - 2):i + 3]: print("  got ", l[:100])
            for l in wl[max(0, i - 2):i + 3]: print("  want", l[:100])
            break
    sys.exit(1)
