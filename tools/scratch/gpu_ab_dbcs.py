import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import refconfig as rc, stringsext_amd as sx
ms = rc.missions(encodings=["big5,,,Cjk", "euc-jp,,,Asian"], chars_min="10")
n = 16 << 30
sc = sx.Scanner(ms, device=0)
d = sc.alloc(n); sc.fill_background(d, 0, n, 0x5EED5EED5EED5EED)
out = []
for k in (0, 1):
    ts = []
    for it in range(4):
        sc.device_runs(k, d, n, stream_parity=0, min_chars=10, count_only=True)
        ts.append(sc.stats().kernel_ms[k])
    out.append(min(ts))
print(os.environ.get("SX_LIB", "new"), "big5 %.2f ms, euc-jp %.2f ms per 16 GiB" % tuple(out))
