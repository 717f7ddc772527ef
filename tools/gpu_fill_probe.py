"""debug: scan time of a buffer that holds a long fill of one lead-range byte (0xF6, 0xE5: format / deleted-entry fills of FAT images)
for the double-byte Missions — the token grid is re-derived by walking back to a byte outside the lead range"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import random
import refconfig as rc, stringsext_amd as sx
enc = sys.argv[1] if len(sys.argv) > 1 else "big5,,,Cjk"
fill = int(sys.argv[2], 16) if len(sys.argv) > 2 else 0xF6
rng = random.Random(1)
ms = rc.missions(encodings=[enc], chars_min="10")
sc = sx.Scanner(ms, device=0)
for mib in [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "1,4,16,64").split(",")]:
    data = rng.randbytes(1 << 20) + bytes([fill]) * (mib << 20) + rng.randbytes(1 << 20)
    sc.reset()
    t0 = time.perf_counter(); res = sc.scan(data, file_id=1); dt = time.perf_counter() - t0
    n = len(res); st = sc.stats(); res.free()
    print(f"{enc} fill 0x{fill:02X} x {mib} MiB: scan {dt * 1e3:.1f} ms (kernel {st.kernel_ms[0]:.2f} ms), {n} findings, replayed {st.replay_bytes} bytes", flush=True)
