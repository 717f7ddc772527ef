import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import refconfig as rc, stringsext_amd as sx
n = int(float(sys.argv[1]) * (1 << 30)) if len(sys.argv) > 1 else 16 << 30
ms = rc.missions(encodings=["utf-8"], chars_min="10")
for sub in (0, 64, 256, 1024):
    sc = sx.Scanner(ms, device=0, subchunk_bytes=sub * 1024)
    d = sc.alloc(n); sc.fill_background(d, 0, n)
    print(f"{n>>30} GiB read probe, pattern {'grid-stride' if sub == 0 else f'sub-chunk {sub}K'}: {sc.read_bandwidth(d, n, 5 if sub == 0 else -5):.0f} GB/s")
    sc.free(d); sc.close()
