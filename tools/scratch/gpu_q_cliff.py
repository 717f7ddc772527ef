"""scratch: what -q > 64 costs (the device replay covers q <= 64; beyond it the host replays): `-e utf-8 -n 10 [-q Q]` on 4 GiB of the synthetic background"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import refconfig as rc, stringsext_amd as sx
n = 4 << 30
for q in (None, "64", "100", "255"):
    kw = dict(encodings=["utf-8"], chars_min="10", unicode_block_filter="African")
    if q: kw["output_line_len"] = q
    sc = sx.Scanner(rc.missions(**kw), device=0)
    d = sc.alloc(n); sc.fill_background(d, 0, n, 12345)
    dts = []
    for it in range(4):
        sc.reset(); t0 = time.perf_counter()
        res = sc.scan_device(d, n, file_id=1); nf = len(res); res.free()
        dts.append(time.perf_counter() - t0)
    print("q", q, f"{min(dts[1:])*1e3:.2f} ms", nf, "findings")
    sc.free(d); sc.close()
