import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
os.environ["SX_WAVE_REPLAY"] = "1"; os.environ["SX_TIMING"] = "1"
import refconfig as rc, stringsext_amd as sx
from test_dbcs import TEXT
txt = TEXT["euc-jp"].encode("euc_jp", "ignore")
ms = rc.missions(encodings=["euc-jp"], chars_min="4", unicode_block_filter="Asian")
f = b"\xa4"
for name, data in (("text", txt * 300), ("fill 30K", txt * 150 + f * 30_000 + txt * 150), ("fill 70K", txt * 150 + f * 70_000 + txt * 150), ("fill 300K", txt * 150 + f * 300_000 + txt * 150),
                   ("fill 300K + A", txt * 150 + f * 300_000 + b"A" + f * 150_001 + txt * 150)):
    sc = sx.Scanner(ms, device=0)
    print("----", name, len(data), flush=True)
    res = sc.scan(data, file_id=1); n = len(res); res.free()
    print("   findings", n, "wave windows", sc.stats().wave_windows, flush=True)
    sc.close()
