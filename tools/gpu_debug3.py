import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import refconfig as rc
import stringsext_amd as sx
import sxo_binding as sxo

m = rc.missions(encodings=["ascii"], chars_min="4")[0]
rng = random.Random(5)
for size in (1 << 20, 1 << 22):
    data = rng.randbytes(size)
    sc = sx.Scanner([m], device=0, subchunk_bytes=65536)
    d = sc.alloc(len(data)); sc.upload(d, data)
    got = sc.device_runs(0, d, len(data), stream_parity=0, min_chars=4)
    want = sxo.runs(m, data, stream_parity=0, min_chars=4)
    print(size, len(got), len(want), got == want, "sorted:", got == sorted(got))
    if got != want:
        for i, (a, b) in enumerate(zip(got, want)):
            if a != b:
                print("first diff", i, got[i-2:i+4], want[i-2:i+4]); break
        sg = set(got); sw = set(want)
        print("missing", len(sw - sg), "extra", len(sg - sw), sorted(sg - sw)[:5])
    sc.close()
