#!/usr/bin/env python3
"""GPU tuning probe: HBM read ceiling, scan-kernel time per mission vs sub-chunk size,
solo vs three concurrent streams.  Writes a plain-text table to stdout."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import refconfig as rc
import stringsext_amd as sx

GIB = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
n = int(GIB * (1 << 30)) // 4096 * 4096
subs = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [16, 64, 256, 1024, 4096]

c3 = rc.missions(encodings=["utf-8", "utf-16le", "utf-16be"], chars_min="10", unicode_block_filter="African")
solo = {"utf8_african": [c3[0]], "utf16le_african": [dict(c3[1], mission_id=0)], "utf16be_african": [dict(c3[2], mission_id=0)],
        "ascii_n4": rc.missions(encodings=["ascii"], chars_min="4"), "utf8_common_n10": rc.missions(encodings=["utf-8"], chars_min="10")}

base = sx.Scanner(c3, device=0)
d = base.alloc(n)
base.fill_background(d, 0, n)
MINC = {"utf8_african_T17": 64, "utf8_common_T17": 64}
solo["utf8_african_T17"] = solo["utf8_african"]
solo["utf8_common_T17"] = solo["utf8_common_n10"]
print(f"buffer {GIB} GiB; read-only streaming probe: {base.read_bandwidth(d, n, 5):.0f} GB/s")
base.close()

print(f"{'mission':18s} {'generic':7s} " + " ".join(f"sub={s:>5d}K" for s in subs) + "   (kernel ms | GB/s)")
for name, ms in solo.items():
    for generic in (False, True):
        row = []
        for s in subs:
            sc = sx.Scanner(ms, device=0, subchunk_bytes=s * 1024, generic_kernels=generic, record_capacity=1 << 22)
            best = 1e9
            for _ in range(3):
                sc.device_runs(0, d, n, 0, MINC.get(name, max(1, min(ms[0]["chars_min_nb"], 64))))
                st = sc.stats()
                best = min(best, st.kernel_ms[0])
            sc.close()
            row.append(f"{best:6.2f}|{n / best / 1e6:5.0f}|h{st.heavy_tiles}")
        print(f"{name:18s} {str(generic):7s} " + " ".join(row))

print("three concurrent streams (C3):")
for s in subs:
    sc = sx.Scanner(c3, device=0, subchunk_bytes=s * 1024)
    best = None
    for _ in range(3):
        sc.reset()
        r = sc.scan_device(d, n, file_id=1)
        st = sc.stats()
        k = [st.kernel_ms[i] for i in range(3)]
        if best is None or max(k) < max(best[0]):
            best = (k, st.device_ms, st.d2h_ms, st.replay_ms, st.total_ms, len(r), st.run_records)
        r.free()
    k = best[0]
    print(f"  sub={s:5d}K kernels {k[0]:.2f}/{k[1]:.2f}/{k[2]:.2f} ms  aggregate {3 * n / max(k) / 1e6:.0f} GB/s | "
          f"device {best[1]:.1f} d2h {best[2]:.1f} replay {best[3]:.1f} total {best[4]:.1f} ms | findings {best[5]} runs {best[6]}")
    sc.close()
