"""debug: count / write pass of the wave kernels for one Mission on the synthetic background (SX_LIB picks the build)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import refconfig as rc, stringsext_amd as sx
enc = sys.argv[1] if len(sys.argv) > 1 else "big5"
n = int(sys.argv[2]) << 30 if len(sys.argv) > 2 else 4 << 30
os.environ["SX_WAVE_REPLAY"] = "1"
ms = rc.missions(encodings=[enc], chars_min=os.environ.get("EXP_N", "10"))   # enc may carry its filter: "big5,,,Cjk"
sc = sx.Scanner(ms, device=0)
d = sc.alloc(n); sc.fill_background(d, 0, n, 0x5EED5EED5EED5EED)
for it in range(3):
    sc.reset()
    t1 = time.perf_counter(); res = sc.scan_device(d, n, file_id=1); t2 = time.perf_counter()
    k = len(res); st = sc.stats(); res.free()
print(f"{os.environ.get('SX_LIB', 'product')} {enc} {n >> 30} GiB: scan {(t2 - t1) * 1e3:.1f} ms; wave windows {st.wave_windows} count {st.wave_count_ms:.2f} write {st.wave_write_ms:.2f} ms; {k} findings", flush=True)
