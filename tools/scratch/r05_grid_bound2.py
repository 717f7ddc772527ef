import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import refconfig as rc, stringsext_amd as sx
os.environ["SX_WAVE_REPLAY"] = "0"
ms = rc.missions(encodings=["euc-kr"], chars_min="4", unicode_block_filter="Kana")
fill, kana = "가".encode("euc_kr"), "あいうえおかきくけこさし".encode("euc_kr")
data = b"x" + (fill * 500 + kana) * 300 + b"\n"
for mode in ("scan_device", "scan"):
    sc = sx.Scanner(ms, device=0, device_replay=True)
    d = sc.alloc(len(data)); sc.upload(d, data)
    for it in range(2):
        sc.reset(); t0 = time.perf_counter()
        res = sc.scan_device(d, len(data), file_id=1) if mode == "scan_device" else sc.scan(data, file_id=1)
        dt = time.perf_counter() - t0; n = len(res); res.free()
        print(mode, it, f"{dt * 1e3:.1f} ms", n, flush=True)
    os.environ["SX_TIMING"] = "1"
    sc.reset(); res = sc.scan_device(d, len(data), file_id=1) if mode == "scan_device" else sc.scan(data, file_id=1); res.free()
    os.environ.pop("SX_TIMING")
    sc.free(d); sc.close()
