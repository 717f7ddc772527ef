"""-r in the wave kernels: shrink the failing input"""
import os, random, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import refconfig as rc, stringsext_amd as sx
import test_wave_core as twc
from test_sharded_gloo import oracle_findings
os.environ["SX_WAVE_REPLAY"] = "1"
m = rc.missions(**twc.SAME_MISSIONS[0])
sc = sx.Scanner(m, device=0, device_replay=True)
def bad(data):
    want = [(a, b, c, d) for a, b, c, d, *_ in oracle_findings([dict(m[0], mission_id=0)], data)]
    sc.reset()
    w0 = sc.stats().wave_windows
    res = sc.scan(data, file_id=1)
    got = [(f["position"], f["precision"], f["s"], f["completes"]) for f in res.findings()]
    res.free()
    return got != want, sc.stats().wave_windows - w0
rng = random.Random(404)
data = twc.same_text(rng, 150_000).encode()
lo, hi = 86016, 95744
print("start", bad(data[lo:hi]))
for step in (4096, 1024, 128):
    while hi - step > 91648 + 128 and bad(data[lo:hi - step])[0]: hi -= step
    while lo + step <= 91648 - 128 and bad(data[lo + step:hi])[0]: lo += step
print("smallest", lo, hi, bad(data[lo:hi]), "failing window at", 91648 - lo)
print(data[lo:hi])
# which bytes matter: replace the windows in front of the failing one by plain text, one at a time
base = bytearray(data[lo:hi])
for w in range(0, (91648 - lo) // 128):
    t = bytearray(base)
    t[w * 128:(w + 1) * 128] = b"z" * 127 + b"\n"
    try:
        bytes(t).decode()
    except UnicodeDecodeError:
        continue
    print("window", w, "replaced:", bad(bytes(t)))
sc.close()
