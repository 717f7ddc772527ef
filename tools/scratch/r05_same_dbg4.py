"""-r: which path gets the 9.7 KB input wrong"""
import os, random, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import refconfig as rc, stringsext_amd as sx
import test_wave_core as twc
from test_sharded_gloo import oracle_findings
m = rc.missions(**twc.SAME_MISSIONS[0])
rng = random.Random(404)
data = twc.same_text(rng, 150_000).encode()
def run(d, env, **kw):
    for k, v in env.items(): os.environ[k] = v
    sc = sx.Scanner(m, device=0, **kw)
    want = [(a, b, c, e) for a, b, c, e, *_ in oracle_findings([dict(m[0], mission_id=0)], d)]
    res = sc.scan(d, file_id=1)
    got = [(f["position"], f["precision"], f["s"], f["completes"]) for f in res.findings()]
    st = sc.stats()
    res.free(); sc.close()
    for k in env: os.environ.pop(k)
    i = next((i for i, (a, b) in enumerate(zip(got, want)) if a != b), None)
    print(len(d), env, kw, "equal" if got == want else f"DIFFERENT at {i}: {got[i][0] if i is not None else None}", "wave windows", st.wave_windows, "fast", st.fast_regions, "general", st.general_regions, flush=True)
small = data[86016:95744]
for d in (small, data):
    run(d, {"SX_WAVE_REPLAY": "0"}, device_replay=True)
    run(d, {"SX_WAVE_REPLAY": "0", "SX_FAST_REPLAY": "0"}, device_replay=True)
    run(d, {"SX_WAVE_REPLAY": "0"}, device_replay=False)
    run(d, {"SX_WAVE_REPLAY": "1"}, device_replay=True)
    run(d, {"SX_WAVE_REPLAY": "1", "SX_WAVE_SAME": "0"}, device_replay=True)
    run(d, {"SX_WAVE_REPLAY": "1"})
