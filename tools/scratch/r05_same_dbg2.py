"""-r in the wave kernels: crafted inputs around the first difference"""
import os, random, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import refconfig as rc, stringsext_amd as sx
import test_wave_core as twc
from test_sharded_gloo import oracle_findings
os.environ["SX_WAVE_REPLAY"] = "1"
m = rc.missions(**twc.SAME_MISSIONS[0])
sc = sx.Scanner(m, device=0)
def check(name, data):
    want = [(a, b, c, d) for a, b, c, d, *_ in oracle_findings([dict(m[0], mission_id=0)], data)]
    sc.reset()
    res = sc.scan(data, file_id=1)
    got = [(f["position"], f["precision"], f["s"], f["completes"]) for f in res.findings()]
    res.free()
    ok = got == want
    print(name, "equal" if ok else "DIFFERENT", len(got), len(want), flush=True)
    if not ok:
        i = next((i for i, (a, b) in enumerate(zip(got, want)) if a != b), min(len(got), len(want)))
        print("   got ", got[i:i + 2], "\n   want", want[i:i + 2])
    return ok
pad = ("x" * 127 + "\n") * 8   # windows of plain text in front (the host does the first ones)
d0, d1 = "изелаиамеайга", "ршутъютчъф"
for k in range(1, 20):
    check(f"D0 x {k} then D1", (pad + "\x01" + d0[:1] * k + d1 + "9e8vvrww\x01" + pad).encode())
for k in (0, 1, 5, 13, 14, 30, 60, 100, 126):
    check(f"shift {k}", (pad + "y" * k + "\x01" + d0 + d1 + "9e8vvrww\x01" + pad).encode())
check("after a rejected 3-byte char", (pad + "uigoi0w1dtl01йア" + d0 + d1 + "9e8vvrww\x01" + pad).encode())
check("after a rejected 3-byte char, leftover", (pad[:-14] + "uigoi0w1dtl01йア" + d0 + d1 + "9e8vvrww\x01" + pad).encode())
check("after a rejected ascii char, leftover", (pad[:-14] + "uigoi0w1dtl01й\x01" + d0 + d1 + "9e8vvrww\x01" + pad).encode())
check("leftover without й", (pad[:-14] + "uigoi0w1dtl01ア" + d0 + d1 + "9e8vvrww\x01" + pad).encode())
check("leftover, й, no reject", (pad[:-14] + "uigoi0w1dtl01й" + d0 + d1 + "9e8vvrww\x01" + pad).encode())
rng = random.Random(404)
data = twc.same_text(rng, 150_000).encode()
for lo, hi in ((91648 - 1024, 91648 + 1024), (91648 - 128, 91648 + 256), (91648 - 4096 - 1536, 91648 + 4096)):
    check(f"the input's bytes [{lo}, {hi})", data[lo:hi])
sc.close()
