"""How do device->host copies travel on this stack?  Times a 256 MiB hipMemcpyAsync into pinned memory next to a long
kernel on another stream (does the copy slow the kernel down / wait for it?), for the environment it is started with.
usage: [HSA_ENABLE_SDMA=1] python tools/gpu_copy_probe.py"""
import os, time, torch
print({k: v for k, v in os.environ.items() if any(t in k for t in ("SDMA", "HSA_", "ROC", "GPU_", "HIP_", "AMD_"))})
dev = torch.device("cuda:0")
n = 256 << 20
src = torch.empty(n, dtype=torch.uint8, device=dev).random_()
dst = torch.empty(n, dtype=torch.uint8).pin_memory()
a = torch.randn(8192, 8192, device=dev)
s_copy, s_comp = torch.cuda.Stream(), torch.cuda.Stream()
def timed(fn, stream):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(stream):
        e0.record(); fn(); e1.record()
    return e0, e1
for it in range(3):
    torch.cuda.synchronize()
    e0, e1 = timed(lambda: dst.copy_(src, non_blocking=True), s_copy)
    torch.cuda.synchronize(); t_copy = e0.elapsed_time(e1)
    k0, k1 = timed(lambda: [a @ a for _ in range(4)], s_comp)
    torch.cuda.synchronize(); t_k = k0.elapsed_time(k1)
    k0, k1 = timed(lambda: [a @ a for _ in range(4)], s_comp)
    e0, e1 = timed(lambda: dst.copy_(src, non_blocking=True), s_copy)
    torch.cuda.synchronize()
    print(f"copy alone {t_copy:.2f} ms ({n / t_copy / 1e6:.1f} GB/s), kernels alone {t_k:.2f} ms; together: copy {e0.elapsed_time(e1):.2f} ms, kernels {k0.elapsed_time(k1):.2f} ms")
# the library's pattern: the copy WAITS for an event of the compute stream, more kernels follow on the compute stream
for it in range(3):
    torch.cuda.synchronize()
    with torch.cuda.stream(s_comp):
        b = a @ a
        ev = torch.cuda.Event(); ev.record()
    s_copy.wait_event(ev)
    e0, e1 = timed(lambda: dst.copy_(src, non_blocking=True), s_copy)
    k0, k1 = timed(lambda: [a @ a for _ in range(4)], s_comp)
    torch.cuda.synchronize()
    print(f"copy behind an event of the compute stream: copy {e0.elapsed_time(e1):.2f} ms, kernels after it {k0.elapsed_time(k1):.2f} ms")
