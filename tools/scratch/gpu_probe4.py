"""Which kinds of work in a second stream make progress while one long kernel runs in the first?"""
import time, torch
dev = torch.device("cuda:0")
big = torch.empty(48 << 30, dtype=torch.uint8, device=dev)
keys = torch.randint(0, 1 << 60, (4_500_000,), dtype=torch.int64, device=dev)
small = torch.zeros(4_500_000, dtype=torch.float32, device=dev)
sa = torch.cuda.Stream(device=dev); sb = torch.cuda.Stream(device=dev)
def trial(name, fn):
    torch.cuda.synchronize()
    fn(); torch.cuda.synchronize()           # warm
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); alone = (time.perf_counter() - t0) * 1e3
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(sa):
        e0.record(); big.add_(1); e1.record()   # one ~17 ms kernel
    time.sleep(0.003)
    t0 = time.perf_counter()
    with torch.cuda.stream(sb):
        fn()
    sb.synchronize()
    busy = (time.perf_counter() - t0) * 1e3
    e1.synchronize()
    print(f"{name:28s} alone {alone:7.2f} ms   next to the long kernel {busy:7.2f} ms   (long kernel {e0.elapsed_time(e1):6.2f} ms)", flush=True)
trial("elementwise x10", lambda: [small.add_(1.0) for _ in range(10)])
trial("memset x10", lambda: [small.zero_() for _ in range(10)])
trial("cumsum (look-back scan)", lambda: torch.cumsum(small, 0))
trial("sort int64 (radix)", lambda: torch.sort(keys))
trial("masked_select (partition)", lambda: torch.masked_select(small, small >= 0))
