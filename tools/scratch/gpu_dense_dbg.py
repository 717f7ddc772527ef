"""debug helper: the dense-strings GPU test case that fails, with the first differing lines"""
import os, random, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import refconfig as rc, sxo_binding as sxo
from product_harness import run_cli_product
from test_gpu_parity import dense
max_gap = int(sys.argv[1]); opts = eval(sys.argv[2])
rng = random.Random(max_gap * 7 + len(opts))
data = dense(rng, 3_000_000 + rng.randrange(5000), max_gap, "abcdefghij XYZ019_-éжЖдяבשλ€")
ms = rc.missions(encodings=["utf-8"], **opts)
want = sxo.run_cli(ms, [data], radix="x")
for chunk in (None, 1 << 20):
    got = run_cli_product(ms, [data], radix="x", chunk_bytes=chunk, device=0, device_replay=True)
    print("chunk", chunk, got == want)
    if got != want:
        gl, wl = got.split(b"\n"), want.split(b"\n")
        print("lines", len(gl), len(wl))
        for i, (a, b) in enumerate(zip(gl, wl)):
            if a != b:
                for l in gl[max(0, i - 3):i + 3]: print("  got ", l[:120])
                for l in wl[max(0, i - 3):i + 3]: print("  want", l[:120])
                break
