"""scratch: device runs vs oracle runs for the tail of the gbk fill case"""
import os, random, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_dbcs import soup, ALL, TEXT, CODEC
import refconfig as rc, sxo_binding as sxo
from test_gpu_parity import device_runs
enc = "gbk"
rng = random.Random(zlib.crc32(enc.encode()) + 7)
txt = TEXT[enc].encode(CODEC[enc], "ignore")
m = rc.missions(encodings=[enc], chars_min="4", unicode_block_filter=ALL)[0]
f = bytes([0xF6]); odd = 0
data = (soup(enc, rng, 3000) + f * (40_000 + odd) + b"A" + f * (70_001 + odd) + txt * 5 + f * (9000 + odd) + b"\x8f" + f * 5000 +
        b"1" + f * (12_345 + odd) + txt[:77] + b"\n")
for k in (0, 127889 - 1 - 1024 * 3 + 3, 136 * 1024 - 1024 * 2 + 1, 136 * 1024 + 1, 137 * 1024 - 55):
    # keep the fill's parity: an odd number of f6 in front of the text
    seg = data[k:]
    for sub in (1024, 4096):
        got, mc = device_runs(m, seg, subchunk=sub)
        want = sxo.runs(m, seg, min_chars=mc)
        print(k, sub, "OK" if got == want else "DIFF", [(a + k, b + k, c) for a, b, c in got[-2:]], [(a + k, b + k, c) for a, b, c in want[-2:]])
