"""PCIe-inclusive rate: a file in the page cache -> pinned host -> HBM -> scan (sx_scan_file), vs chunk size."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import refconfig as rc, stringsext_amd as sx, sxo_binding as sxo
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(gib * (1 << 30))
path = "/dev/shm/sx_ingest.bin" if os.path.isdir("/dev/shm") else "/tmp/sx_ingest.bin"
with open(path, "wb") as f:
    step = 256 << 20
    for off in range(0, n, step):
        f.write(sxo.background(off, min(step, n - off)))
ms = rc.missions(encodings=["utf-8", "utf-16le", "utf-16be"], chars_min="10", unicode_block_filter="African")
for chunk_mib in (64, 256, 1024):
    sc = sx.Scanner(ms, device=0)
    best = None
    for rep in range(3):
        sc.reset()
        t0 = time.perf_counter()
        parts = sc.scan_file(path, chunk_bytes=chunk_mib << 20)
        dt = time.perf_counter() - t0
        nf = sum(len(p) for p in parts)
        for p in parts:
            p.free()
        best = dt if best is None else min(best, dt)
    print(f"chunk {chunk_mib:5d} MiB: {gib / best:7.1f} GiB/s ({best * 1e3:.0f} ms for {gib} GiB, {nf} findings)", flush=True)
    sc.close()
os.unlink(path)
