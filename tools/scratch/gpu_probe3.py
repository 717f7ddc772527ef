"""How soon does a small kernel in a second stream run while a long HBM-bound kernel is busy?"""
import time, torch
dev = torch.device("cuda:0")
big = torch.empty(8 << 30, dtype=torch.uint8, device=dev)
small = torch.zeros(1 << 20, dtype=torch.float32, device=dev)
sa = torch.cuda.Stream(device=dev, priority=0)
for prio in (0, -1):
    sb = torch.cuda.Stream(device=dev, priority=prio)
    torch.cuda.synchronize()
    for rep in range(3):
        e0, e1, f0, f1 = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        with torch.cuda.stream(sa):
            e0.record()
            for _ in range(4):
                big.add_(1)          # 4 x (8 GiB read + 8 GiB write)
            e1.record()
        time.sleep(0.002)
        t0 = time.perf_counter()
        with torch.cuda.stream(sb):
            f0.record()
            for _ in range(10):
                small.add_(1.0)      # ten tiny dependent kernels
            f1.record()
        f1.synchronize()
        t_small = (time.perf_counter() - t0) * 1e3
        e1.synchronize()
        print(f"prio {prio}: long stream {e0.elapsed_time(e1):7.2f} ms; ten small kernels: host-visible {t_small:7.2f} ms, "
              f"device {f0.elapsed_time(f1):7.2f} ms", flush=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    small.add_(1.0)
torch.cuda.synchronize()
print(f"idle GPU: ten small kernels {1e3*(time.perf_counter()-t0):.2f} ms")
