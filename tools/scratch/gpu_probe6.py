"""UTF-8 scan kernel: how much of its time is the slow path?  Same kernel on random bytes (5 % of the
tiles hold a candidate) and on bytes without any accepted stretch (pure fast path)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import refconfig as rc, stringsext_amd as sx
n = 32 << 30
m8 = rc.missions(encodings=["utf-8"], chars_min="10", unicode_block_filter="African")[0]
m16 = rc.missions(encodings=["utf-16le"], chars_min="10", unicode_block_filter="African")[0]
import ctypes
buf = torch.empty(n, dtype=torch.uint8, device="cuda:0")
d = ctypes.c_void_p(buf.data_ptr())
for name, m in (("utf-8", m8), ("utf-16le", m16)):
    sc = sx.Scanner([m], device=0)
    for label, fill in (("random", None), ("all 0x00", 0), ("all 0xFF", 255), ("random & 0x1F (controls only)", "ctl")):
        if fill is None:
            sc.fill_background(d, 0, n)
        elif fill == "ctl":
            sc.fill_background(d, 0, n); buf.bitwise_and_(0x1F)
        else:
            buf.fill_(fill)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(4):
            sc.device_runs(0, d, n, stream_parity=0, min_chars=10, count_only=True)
            best = min(best, sc.stats().kernel_ms[0])
        print(f"{name:9s} {label:32s}: {best:7.3f} ms  {n / best / 1e6:7.1f} GB/s", flush=True)
    sc.close()
