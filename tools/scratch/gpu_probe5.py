"""Read bandwidth of access patterns a scan kernel could use (no classification work)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import refconfig as rc, stringsext_amd as sx
n = 32 << 30
ms = rc.missions(encodings=["utf-8"], chars_min="10")
sc0 = sx.Scanner(ms, device=0)
d = sc0.alloc(n); sc0.fill_background(d, 0, n)
print(f"grid-stride: {sc0.read_bandwidth(d, n, 5):.0f} GB/s", flush=True)
for sub in (16, 64, 256, 1024):
    sc = sx.Scanner(ms, device=0, subchunk_bytes=sub * 1024)
    for mode in (0, 1):
        for loads in (1, 2, 4):
            os.environ["SX_PROBE_LOADS"] = str(loads); os.environ["SX_PROBE_MODE"] = str(mode)
            print(f"sub {sub:5d}K mode {mode} loads {loads}: {sc.read_bandwidth(d, n, -4):.0f} GB/s", flush=True)
    sc.close()
sc0.free(d); sc0.close()
