"""Read probe with private sub-chunks whose wavefronts start at rotated offsets (round 5).  usage: r05_rot_probe.py [GIB]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import refconfig as rc, stringsext_amd as sx
GIB = float(sys.argv[1]) if len(sys.argv) > 1 else 32.0
n = int(GIB * (1 << 30)) // 4096 * 4096
c3 = rc.missions(encodings=["utf-8"], chars_min="10")
base = sx.Scanner(c3, device=0)
d = base.alloc(n); base.fill_background(d, 0, n)
print(f"buffer {GIB} GiB; grid-stride read probe {base.read_bandwidth(d, n, 5):.0f} GB/s", flush=True)
base.close()
for s in (256, 128, 512, 1024):
    for rot in (0, 1, 3, 37, 97, 101):
        os.environ["SX_PROBE_ROT"] = str(rot)
        sc = sx.Scanner(c3, device=0, subchunk_bytes=s * 1024)
        print(f"  sub {s:5d} KiB rot {rot:4d}: {sc.read_bandwidth(d, n, -5):.0f} GB/s", flush=True)
        sc.close()
