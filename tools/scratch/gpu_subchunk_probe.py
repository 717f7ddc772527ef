"""Does the sub-chunk SIZE matter beyond its magnitude?  Wavefront w streams bytes [w * sub, (w + 1) * sub): with sub = 256 KiB all
wavefronts that are equally far into their sub-chunk read addresses that are equal modulo 256 KiB — the same memory channel if the
channel interleave divides that.  A size that is no multiple of the interleave period spreads them.  usage: gpu_subchunk_probe.py [GIB]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import refconfig as rc, stringsext_amd as sx
n = (int(sys.argv[1]) if len(sys.argv) > 1 else 32) << 30
ms = rc.missions(encodings=["utf-8"], chars_min="10")
sc0 = sx.Scanner(ms, device=0)
d = sc0.alloc(n); sc0.fill_background(d, 0, n)
print(f"grid-stride: {sc0.read_bandwidth(d, n, 5):.0f} GB/s", flush=True)
os.environ.pop("SX_PROBE_LOADS", None); os.environ.pop("SX_PROBE_MODE", None)
for sub in (256, 255, 257, 258, 260, 264, 272, 288, 320, 129, 136, 65, 72, 513):
    sc = sx.Scanner(ms, device=0, subchunk_bytes=sub * 1024)
    print(f"sub {sub:5d} KiB: {sc.read_bandwidth(d, n, -4):.0f} GB/s", flush=True)
    sc.close()
sc0.free(d); sc0.close()
