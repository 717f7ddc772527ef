"""debug: where a step of `bench.py --workload c1` goes (python side vs library), and the slab count"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import refconfig as rc, stringsext_amd as sx
ms = rc.missions(encodings=["ascii"], chars_min="4")
n = 1 << 30
sc = sx.Scanner(ms, device=0)
d = sc.alloc(n); sc.fill_background(d, 0, n, 0x5EED5EED5EED5EED)
for slabs in (None, "4", "16", "32"):
    if slabs: os.environ["SX_WAVE_SLABS"] = slabs
    else: os.environ.pop("SX_WAVE_SLABS", None)
    ts = []
    for it in range(8):
        t0 = time.perf_counter(); sc.reset(); t1 = time.perf_counter()
        res = sc.scan_device(d, n, file_id=1); t2 = time.perf_counter()
        k = len(res); st = sc.stats(); t3 = time.perf_counter()
        res.free(); t4 = time.perf_counter()
        ts.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3, st.total_ms, st.wave_count_ms, st.wave_write_ms))
    a = ts[-1]
    print(f"slabs {slabs}: reset {a[0]*1e3:.3f} scan {a[1]*1e3:.3f} (library total {a[4]:.3f}; wave count {a[5]:.2f} write {a[6]:.2f}) len+stats {a[2]*1e3:.3f} free {a[3]*1e3:.3f} ms; {k} findings", flush=True)
