"""dbcs_sync_before with and without the grid bound: time of a scan whose regions lie inside megabytes of lead-range bytes"""
import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import refconfig as rc, stringsext_amd as sx
os.environ["SX_WAVE_REPLAY"] = "0"
ms = rc.missions(encodings=["euc-kr"], chars_min="4", unicode_block_filter="Kana")
fill, kana = "가".encode("euc_kr"), "あいうえおかきくけこさし".encode("euc_kr")
for reps in (1000, 10000):
    data = b"x" + (fill * 500 + kana) * reps + b"\n"
    for env in ({}, {"SX_NO_GRID_BOUND": "1"}):
        for k, v in env.items(): os.environ[k] = v
        sc = sx.Scanner(ms, device=0, device_replay=True)
        d = sc.alloc(len(data)); sc.upload(d, data)
        dts = []
        for it in range(3):
            sc.reset(); t0 = time.perf_counter(); res = sc.scan_device(d, len(data), file_id=1); dts.append(time.perf_counter() - t0); n = len(res); res.free()
        print(f"{len(data) / 2**20:.1f} MiB, {reps} regions, {env or 'grid bound'}: {min(dts) * 1e3:.1f} ms, {n} findings", flush=True)
        sc.free(d); sc.close()
        for k in env: os.environ.pop(k)
