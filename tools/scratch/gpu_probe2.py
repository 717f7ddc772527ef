"""Is a scan kernel bound by HBM or by its instruction stream?  Same kernel, same number of
wavefronts and bytes per wavefront, once streaming through a large HBM-resident buffer and once
re-reading a buffer small enough to stay in the 256 MiB infinity cache."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import refconfig as rc, stringsext_amd as sx

def probe(name, m, n, sub_kib, reps=6):
    sc = sx.Scanner([m], device=0, subchunk_bytes=sub_kib * 1024)
    d = sc.alloc(n); sc.fill_background(d, 0, n)
    best = 1e9
    for _ in range(reps):
        sc.device_runs(0, d, n, stream_parity=0, min_chars=10)
        best = min(best, sc.stats().kernel_ms[0])
    print(f"{name:14s} {n/2**20:8.0f} MiB sub {sub_kib:4d}K waves {n//(sub_kib*1024):6d}: {best:8.3f} ms  {n/best/1e6:8.1f} GB/s", flush=True)
    sc.free(d); sc.close()

ms = {
  "utf8_african": rc.missions(encodings=["utf-8"], chars_min="10", unicode_block_filter="African")[0],
  "utf16le_afr": rc.missions(encodings=["utf-16le"], chars_min="10", unicode_block_filter="African")[0],
  "utf8_common": rc.missions(encodings=["utf-8"], chars_min="10")[0],
}
for name, m in ms.items():
    for n, sub in ((96 << 20, 16), (96 << 20, 64), (8 << 30, 16), (8 << 30, 64), (8 << 30, 256), (32 << 30, 256)):
        probe(name, m, n, sub)
