import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import refconfig as rc, stringsext_amd as sx, sxo_binding as sxo
from product_harness import run_cli_product
from test_host_logic import soup, synth
from test_gpu_parity import dense
ENCS = ["utf-8", "ascii", "utf-16le", "utf-16be", "koi8-r", "ibm866", "windows-1252", "iso-8859-5", "x-user-defined"]
AFS = [None, "All", "All-Ctrl", "All-Ctrl+Wsp", "None", "Wsp", "0x7ffffffe000000007ffffffe00000000"]
UBFS = [None, "African", "All", "Common", "Cyrillic", "Latin", "Asian", "Uncommon", "None", "Hebrew", "Cjk"]
case_seed = int(sys.argv[1])
r = random.Random(case_seed)
encs = []
for _ in range(r.randrange(1, 4)):
    e = r.choice(ENCS)
    if r.random() < 0.3:
        e += "," + r.choice(["", "2", "5", "12"]) + "," + (r.choice(AFS) or "") + "," + (r.choice(UBFS) or "")
    encs.append(e)
kw = dict(encodings=encs, chars_min=r.choice([None, "1", "2", "4", "7", "10", "20", "70"]),
          output_line_len=r.choice([None, None, "6", "8", "10", "30", "64", "100"]),
          ascii_filter=r.choice(AFS), unicode_block_filter=r.choice(UBFS),
          grep_char=r.choice([None, None, None, "47", "0x65", "32"]), same_unicode_block=r.random() < 0.2,
          counter_offset=r.choice([None, None, "1000", "0x10"]))
ms = rc.missions(**kw)
kind = r.choice(["synth", "synth_dense", "soup", "dense", "text", "multi"])
size = r.choice([5000, 70_000, 300_000, 1_200_000])
if kind == "synth": files = [synth(r, size, 1 / 500)]
elif kind == "synth_dense": files = [synth(r, size, 1 / 60)]
elif kind == "soup": files = [soup(r, min(size, 200_000))]
elif kind == "dense": files = [dense(r, size, r.choice([2, 20, 200]), "abcdefgh XYZ019_-éжЖдяבשλ€😀")]
elif kind == "text": files = [("The quick brown fox — Ünïcödé ßtring, доброе утро, שלום עולם. " * (size // 60 + 1)).encode(r.choice(["utf-8", "utf-16-le", "koi8-r"]), errors="replace")[:size]]
else: files = [synth(r, r.randrange(1, 20000), 1 / 100) for _ in range(r.randrange(2, 6))] + [b""]
chunk = r.choice([None, None, 4096, 16384, 65536])
flush = r.random() < 0.3
sub = r.choice([0, 0, 1024, 4096])
replay = r.choice([None, None, True, False])
print(kw, kind, size, chunk, flush)
want = sxo.run_cli(ms, files, radix="x", flush_at_eof=flush)
got = run_cli_product(ms, files, radix="x", chunk_bytes=chunk, device=0, flush_at_eof=flush, device_replay=(None if len(sys.argv) < 3 else sys.argv[2] == "dev"))
print("gpu path equal:", got == want)
if got != want:
    gl, wl = got.split(b"\n"), want.split(b"\n")
    for i, (a, b) in enumerate(zip(gl, wl)):
        if a != b:
            print(i, a, b); print(gl[i-1:i+3]); print(wl[i-1:i+3]); break
    d = files[0]
    print(d[0x43000-24:0x43000+24])
d = files[0]
off = 0x81d4 - 1000
print("window start", off, (off - off // 4096 * 4096) % 12)
print(d[off-14:off+26])
print(d[off-14:off+26].decode("iso-8859-5").encode("unicode_escape"))
m = ms[0]
print(m)
