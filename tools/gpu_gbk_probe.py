import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import stringsext_amd as sx
for enc, ubf in (("gbk", "Cjk"), ("big5", "Cjk")):
    ms = sx.missions_from_flags(encodings=[enc + ",,," + ubf], chars_min="10")
    sc = sx.Scanner(ms, device=0)
    n = 16 << 30
    d = sc.alloc(n); sc.fill_background(d, 0, n, 0x5EED5EED5EED5EED)
    for it in range(3):
        sc.reset(); t0 = time.perf_counter(); res = sc.scan_device(d, n, file_id=1); dt = time.perf_counter() - t0; nf = len(res); res.free()
    st = sc.stats() if hasattr(sc, "stats") else None
    print(enc, f"16 GiB: {dt*1e3:.1f} ms = {16/dt:.1f} GiB/s, {nf} findings", st)
    sc.free(d); sc.close()
