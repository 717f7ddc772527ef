"""Scan-kernel time of the three headline Missions, solo, by sub-chunk size (round 5).  usage: r05_sub_probe.py [GIB] [subs,...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import refconfig as rc, stringsext_amd as sx
GIB = float(sys.argv[1]) if len(sys.argv) > 1 else 16.0
n = int(GIB * (1 << 30)) // 4096 * 4096
subs = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [16, 32, 64, 128, 256]
c3 = rc.missions(encodings=["utf-8", "utf-16le", "utf-16be"], chars_min="10", unicode_block_filter="African")
solo = {"utf8": [c3[0]], "utf16le": [dict(c3[1], mission_id=0)], "utf16be": [dict(c3[2], mission_id=0)]}
base = sx.Scanner(c3, device=0)
d = base.alloc(n); base.fill_background(d, 0, n)
print(f"buffer {GIB} GiB; grid-stride read probe {base.read_bandwidth(d, n, 5):.0f} GB/s", flush=True)
for s in subs:
    sc = sx.Scanner(c3, device=0, subchunk_bytes=s * 1024)
    print(f"  read probe, private sub-chunks of {s} KiB: {sc.read_bandwidth(d, n, -4):.0f} GB/s", flush=True)
    sc.close()
base.close()
for name, ms in solo.items():
    for s in subs:
        sc = sx.Scanner(ms, device=0, subchunk_bytes=s * 1024, record_capacity=1 << 22)
        ts = []
        for _ in range(4):
            sc.device_runs(0, d, n, 0, 10)
            ts.append(sc.stats().kernel_ms[0])
        sc.close()
        best = min(ts)
        print(f"{name:8s} sub {s:4d} KiB: {best:7.3f} ms = {n / best / 1e6:5.0f} GB/s   (all: {' '.join(f'{t:.2f}' for t in ts)})", flush=True)
