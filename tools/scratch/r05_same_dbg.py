"""-r in the wave kernels: where the device's findings leave the oracle's (first difference, with the input around it), under a few switches"""
import os, random, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import refconfig as rc, stringsext_amd as sx
import test_wave_core as twc
from test_sharded_gloo import oracle_findings
os.environ["SX_WAVE_REPLAY"] = "1"
rng = random.Random(404)
kw = twc.SAME_MISSIONS[0]
m = rc.missions(**kw)
data = twc.same_text(rng, 150_000).encode()
want = oracle_findings([dict(m[0], mission_id=0)], data)
def product(env):
    for k, v in env.items(): os.environ[k] = v
    sc = sx.Scanner(m, device=0)
    res = sc.scan(data, file_id=1)
    got = [(f["position"], f["precision"], f["s"], f["completes"]) for f in res.findings()]
    st = sc.stats()
    res.free(); sc.close()
    for k in env: os.environ.pop(k)
    return got, st
for env in ({}, {"SX_WAVE_DESC": "0"}, {"SX_WAVE_LUT": "1"}, {"SX_WAVE_BATCHES": "1"}, {"SX_WAVE_SAME": "0"}, {"SX_WAVE_REPAIR": "0"}):
    got, st = product(env)
    w2 = [(a, b, c, d) for a, b, c, d, *_ in want]
    ok = got == w2
    print(env, "equal" if ok else "DIFFERENT", len(got), len(w2), "wave windows", st.wave_windows, "repairs", st.wave_repairs, flush=True)
    if not ok:
        i = next((i for i, (a, b) in enumerate(zip(got, w2)) if a != b), min(len(got), len(w2)))
        print("  first diff at", i, "\n  got ", got[max(0, i - 1):i + 3], "\n  want", w2[max(0, i - 1):i + 3])
        p = w2[i][0] if i < len(w2) else got[i][0]
        # the text of the finding in the input
        t = w2[i][2].encode() if i < len(w2) else b""
        at = data.find(t, max(0, p - 4096)) if t else p
        print("  input offset of the wanted string", at, "window", at // 128, "offset in slice", at % 4096, "\n  ", data[max(0, at - 40):at + 80])
