"""scratch: one UTF-16 wave-path case against the oracle, first differing line with context"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import refconfig as rc, sxo_binding as sxo
from product_harness import run_cli_product
from test_wave_core import UTF16_MISSIONS, utf16_soup
ui, chunk, batches = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
flags = UTF16_MISSIONS[ui]
be = flags["encodings"][0].endswith("be")
ms = rc.missions(**dict(flags, encodings=flags["encodings"] + ["utf-8"]))
rng = random.Random(8000 + ui)
from test_wave_core import text_lines
text = text_lines(rng, 150_000).decode("latin-1").encode("utf-16-be" if be else "utf-16-le")
datas = [("text", text), ("soup", utf16_soup(rng, 150_000, be)), ("random", rng.randbytes(200_000)),
         ("astral", ("a\U0001F600b\U00020000\U0001F601cd 中" * 9000).encode("utf-16-be" if be else "utf-16-le")),
         ("high surrogates", utf16_soup(rng, 100_000, be, (40, 10, 14, 10, 10, 4)))]
name, data = datas[int(sys.argv[4])]
os.environ["SX_WAVE_REPLAY"] = sys.argv[5] if len(sys.argv) > 5 else "1"
os.environ["SX_WAVE_BATCHES"] = batches
want = sxo.run_cli(ms, [data], radix="x").split(b"\n")
got = run_cli_product(ms, [data], radix="x", device=0, chunk_bytes=chunk).split(b"\n")
print(name, len(data), "lines", len(got), len(want))
for i, (a, b) in enumerate(zip(got, want)):
    if a != b:
        for j in range(max(0, i - 3), min(len(want), i + 4)):
            print(j, "got ", got[j] if j < len(got) else None); print(j, "want", want[j])
        break
else:
    print("equal" if len(got) == len(want) else "length differs")
