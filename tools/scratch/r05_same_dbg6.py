"""windows-1253 -r on random bytes: which switch matters"""
import os, random, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import refconfig as rc, stringsext_amd as sx
import test_wave_core as twc
from test_sharded_gloo import oracle_findings
os.environ["SX_WAVE_REPLAY"] = "1"
rng = random.Random(404)
for ki, kw in enumerate(twc.SAME_MISSIONS + twc.SAME_UTF16):
    codec = kw["encodings"][0]
    enc = lambda t: t.encode(codec, errors="replace" if not codec.startswith("utf-") else "strict")
    datas = [enc(twc.same_text(rng, 150_000)), enc(twc.same_text(rng, 60_000, runs=(1, 7, 30, 64, 65, 130))), enc(twc.russian(rng, 100_000)), rng.randbytes(100_000)]
    if ki == 7: break
m = rc.missions(**kw)
data = datas[3]
want = [(a, b, c, d) for a, b, c, d, *_ in oracle_findings([dict(m[0], mission_id=0)], data)]
for env in ({}, {"SX_WAVE_DESC": "0"}, {"SX_WAVE_BATCHES": "2"}, {"SX_WAVE_DESC_CAP": "200"}, {"SX_WAVE_LUT": "1"}):
    for k, v in env.items(): os.environ[k] = v
    sc = sx.Scanner(m, device=0, device_replay=True)
    res = sc.scan(data, file_id=1)
    got = []
    for v, n, arena in res.segments():
        got += [(v[i].position, sx.PRECISION[v[i].precision], arena[v[i].str_off:v[i].str_off + v[i].str_len].decode("utf-8", "replace"), bool(v[i].completes_previous)) for i in range(n)]
    st = sc.stats()
    res.free(); sc.close()
    for k in env: os.environ.pop(k)
    i = next((i for i, (a, b) in enumerate(zip(got, want)) if a != b), None)
    if i is not None:
        print("positions got ", [g[0] for g in got[i - 3:i + 12]])
        print("positions want", [g[0] for g in want[i - 3:i + 12]])
        print("wrong (want pos, got pos, window in its wavefront's batch):", [(w[0], g[0], (w[0] // 16 - (w[0] // 16 // 60 * 60 - (4 if w[0] // 16 >= 60 else 0))) % 64) for g, w in zip(got, want) if g != w][:60])
        sg = set(got); miss = [w for w in want if w not in sg]
        print("wanted findings that are nowhere in the result:", len(miss), "first", miss[:2], "; equal again from", next((j for j in range(i, len(want)) if got[j:] == want[j:]), None))
    print(env, "equal" if got == want else f"DIFFERENT at {i}: got {got[i:i+2]} want {want[i:i+2]}", len(got), len(want), "wave windows", st.wave_windows, "repairs", st.wave_repairs, "desc overflows", st.wave_desc_overflows, flush=True)
