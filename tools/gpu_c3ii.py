"""C3(ii): the 'disk image' (records planted every 64 KiB in 3 encodings) — whole-job time when all missions have findings."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import refconfig as rc, stringsext_amd as sx
from test_gpu_baseline_configs import planted_image
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 2
n = int(gib * (1 << 30))
every = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
ms = rc.missions(encodings=["utf-8", "utf-16le", "utf-16be"], chars_min="10", unicode_block_filter="African")
img = planted_image(n, 3, every=every)[:n]
sc = sx.Scanner(ms, device=0)
d = sc.alloc(n); sc.upload(d, img)
for rep in range(3):
    sc.reset()
    t0 = time.perf_counter()
    res = sc.scan_device(d, n, file_id=1)
    dt = time.perf_counter() - t0
    per = {}
    print(f"{gib} GiB, record every {every} B: {1e3 * dt:.1f} ms = {gib / dt:.1f} GiB/s, {len(res)} findings, {len(res.segments())} segment(s)", flush=True)
    res.free()
sc.free(d); sc.close()
