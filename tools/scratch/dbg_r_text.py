"""scratch: where the time goes for `-e utf-8 -r` on text (SX_TIMING=1)"""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import refconfig as rc, stringsext_amd as sx
from test_wave_core import text_lines
data = text_lines(random.Random(1), 64 << 20)
ms = rc.missions(encodings=["utf-8"], chars_min="10", same_unicode_block=True)
sc = sx.Scanner(ms, device=0)
d = sc.alloc(len(data)); sc.upload(d, data)
for it in range(3):
    sc.reset(); t0 = time.perf_counter()
    if it == 2: os.environ["SX_TIMING"] = "1"
    res = sc.scan_device(d, len(data), file_id=1); n = len(res); res.free()
    print(it, (time.perf_counter() - t0) * 1e3, "ms", n, file=sys.stderr)
st = sc.stats()
print({k: getattr(st, k) for k in dir(st) if not k.startswith("_") and isinstance(getattr(st, k), (int, float))}, file=sys.stderr)
