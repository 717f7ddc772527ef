"""debug: one Mission through the wave kernels on the synthetic background at a given size.  usage: gpu_wave_size.py ENC GIB [flags...]"""
import os, sys, time, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import refconfig as rc, stringsext_amd as sx
enc, gib = sys.argv[1], float(sys.argv[2])
ms = rc.missions(encodings=enc.split("+"), chars_min="10")
n = int(gib * (1 << 30)) // 4096 * 4096
sc = sx.Scanner(ms, device=0)
d = sc.alloc(n)
sc.fill_background(d, 0, n, 0x5EED5EED5EED5EED)
for it in range(3):
    sc.reset(); t0 = time.perf_counter()
    res = sc.scan_device(d, n, file_id=1)
    dt = time.perf_counter() - t0
    st = sc.stats()
    print(enc, gib, "GiB:", f"{dt*1e3:.1f} ms", len(res), "findings, wave windows", st.wave_windows, f"count {st.wave_count_ms:.2f} write {st.wave_write_ms:.2f} ms", flush=True)
    res.free()
sc.free(d); sc.close()
