import os, random, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import refconfig as rc, stringsext_amd as sx
import test_wave_core as twc
os.environ["SX_WAVE_REPLAY"] = "1"; os.environ["SX_WAVE_DBG_WIN"] = "44"
m = rc.missions(**twc.SAME_MISSIONS[0])
rng = random.Random(404)
data = twc.same_text(rng, 150_000).encode()[86016:95744]
sc = sx.Scanner(m, device=0, device_replay=True)
res = sc.scan(data, file_id=1); print(len(res)); res.free(); sc.close()
