import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import refconfig as rc, stringsext_amd as sx, sxo_binding as sxo
m = rc.missions(encodings=["ascii"], chars_min="4")[0]
data = bytearray(b"\x00" * 16384)
data[12275:12341] = b"x" * 66
data[2040:2050] = b"abcdefghij"      # crosses 2048
data[3070:3073] = b"abc"; data[3073:3076]=b"def"   # 3+3 across 3072
data[5110:5120] = b"0123456789"      # ends exactly at 5120
data[6144:6150] = b"qwerty"          # starts exactly at 6144
data = bytes(data)
for sub in (1024, 2048):
    sc = sx.Scanner([m], device=0, subchunk_bytes=sub)
    d = sc.alloc(len(data)); sc.upload(d, data)
    got = sc.device_runs(0, d, len(data), 0, 4)
    want = sxo.runs(m, data, min_chars=4)
    print("sub", sub, "got", got, "want", want, "OK" if got == want else "MISMATCH")
    sc.free(d); sc.close()
