"""scratch: text with -q 100 (lane-per-region path: the wave path covers q <= 64)"""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import refconfig as rc, stringsext_amd as sx, sxo_binding as sxo
from test_wave_core import text_lines
from product_harness import run_cli_product
data = text_lines(random.Random(1), 64 << 20)
for q in ("100", "255"):
    ms = rc.missions(encodings=["utf-8"], chars_min="10", output_line_len=q)
    small = data[:3 << 20]
    print("q", q, "parity on 3 MiB:", run_cli_product(ms, [small], radix="x", device=0, chunk_bytes=1 << 20) == sxo.run_cli(ms, [small], radix="x"))
    sc = sx.Scanner(ms, device=0)
    d = sc.alloc(len(data)); sc.upload(d, data)
    dts = []
    for it in range(4):
        sc.reset(); t0 = time.perf_counter()
        res = sc.scan_device(d, len(data), file_id=1); n = len(res); res.free()
        dts.append(time.perf_counter() - t0)
    print("q", q, f"64 MiB text: {min(dts[1:])*1e3:.1f} ms = {64/1024/min(dts[1:]):.2f} GiB/s", n)
    sc.free(d); sc.close()
