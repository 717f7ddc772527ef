"""Text-like input (lines of 10-120 printable chars ended by \\n): how the whole path behaves when nearly every
window start is crossed by a run.  usage: tools/gpu_text.py [MIB [ENCODING | russian ...]]"""
import os, random, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import refconfig as rc, stringsext_amd as sx
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rng = random.Random(1)
words = [bytes(rng.choice(b"abcdefghijklmnopqrstuvwxyzABCDEFGH0123456789_-./:=") for _ in range(rng.randrange(2, 12))) for _ in range(500)]
lines = []
for _ in range(20000):
    n = rng.randrange(10, 120); l = bytearray()
    while len(l) < n: l += rng.choice(words) + b" "
    lines.append(bytes(l[:n]) + b"\n")
blob = b"".join(lines)
data = (blob * (mib * (1 << 20) // len(blob) + 1))[:mib << 20]
blob16 = blob.decode().encode("utf-16-le")
data16 = (blob16 * (mib * (1 << 20) // len(blob16) + 1))[:mib << 20]
cases = [(dict(encodings=["ascii"], chars_min="4"), data), (dict(encodings=["utf-8"], chars_min="10"), data),
         (dict(encodings=["utf-16le"], chars_min="10"), data16),
         # -r: `-e ascii` (no two accepted characters with different UTF-8 lead bytes exist) and `-e utf-8` / `-e utf-16le` (none in this buffer): the wave path
         (dict(encodings=["ascii"], chars_min="4", same_unicode_block=True), data), (dict(encodings=["utf-8"], chars_min="10", same_unicode_block=True), data),
         (dict(encodings=["utf-16le"], chars_min="10", same_unicode_block=True), data16)]
# -g (round 5: on the wave path — the char of the reference's functional test 2, `:`, which a tenth of these lines hold; and a char none holds)
cases += [(dict(encodings=["ascii"], chars_min="4", grep_char="58"), data), (dict(encodings=["utf-8"], chars_min="10", grep_char="58"), data),
          (dict(encodings=["utf-16le"], chars_min="10", grep_char="58"), data16), (dict(encodings=["utf-8"], chars_min="10", grep_char="63"), data)]
# -r that matters (round 5: in the wave kernels): Russian text — the lead byte of its letters changes inside nearly every word (D0 / D1) —,
# as UTF-8, KOI8-R and UTF-16LE; and the same text without -r beside it
cyr_words = ["".join(rng.choice("абвгдежзийклмнопрстуфхцчшщъыьэюя") for _ in range(rng.randrange(2, 12))) for _ in range(500)]
cl = []
for _ in range(20000):
    n = rng.randrange(10, 120); l = ""
    while len(l) < n: l += rng.choice(cyr_words) + " "
    cl.append(l[:n] + "\n")
cyr = "".join(cl)
def rep(b): return (b * (mib * (1 << 20) // len(b) + 1))[:mib << 20 if len(b) % 2 == 0 else mib << 20]
c8, ck, c16 = cyr.encode("utf-8"), cyr.encode("koi8-r"), cyr.encode("utf-16-le")
c8 = c8 + b" " * (-len(c8) % 4)   # (repeated whole: no character is cut where the copies meet)
cases += [(dict(encodings=["utf-8"], chars_min="4", unicode_block_filter="Cyrillic", same_unicode_block=True, name="russian"), rep(c8)),
          (dict(encodings=["utf-8"], chars_min="4", unicode_block_filter="Cyrillic", name="russian"), rep(c8)),
          (dict(encodings=["koi8-r"], chars_min="4", unicode_block_filter="Cyrillic", same_unicode_block=True, name="russian"), rep(ck)),
          (dict(encodings=["utf-16le"], chars_min="4", unicode_block_filter="Cyrillic", same_unicode_block=True, name="russian"), rep(c16)),
          (dict(encodings=["utf-8"], chars_min="4", unicode_block_filter="Cyrillic", same_unicode_block=True, grep_char="32", name="russian"), rep(c8))]   # -g AND -r
# -q beyond the wave path's 64: text at -q 255 on the lane-per-region path
cases += [(dict(encodings=["ascii"], chars_min="4", output_line_len="255", name="q255"), data), (dict(encodings=["utf-8"], chars_min="10", output_line_len="255", name="q255"), data)]
if len(sys.argv) > 2: cases = [c for c in cases if c[0]["encodings"][0] in sys.argv[2:] or c[0].get("name") in sys.argv[2:]]
for flags, data in cases:
    what = flags.pop("name", "text")
    ms = rc.missions(**flags)
    sc = sx.Scanner(ms, device=0, result_on_device=bool(os.environ.get("RESULT_ON_DEVICE")))   # (RESULT_ON_DEVICE=1: SX_OPT_RESULT_ON_DEVICE — the findings stay in HBM)
    d = sc.alloc(len(data)); sc.upload(d, data)
    dts = []
    for it in range(8):   # the first passes size the pinned pool and the record regions
        sc.reset(); t0 = time.perf_counter()
        res = sc.scan_device(d, len(data), file_id=1)
        dts.append(time.perf_counter() - t0); n = len(res); res.free()
    dt = sorted(dts[2:])[len(dts[2:]) // 2]
    print(flags["encodings"], "-r" if flags.get("same_unicode_block") else "", f"-g {flags['grep_char']}" if flags.get("grep_char") else "", f"{mib} MiB {what}: {dt*1e3:.1f} ms = {mib/1024/dt:.2f} GiB/s (median of 6; min {min(dts)*1e3:.1f}, max {max(dts[2:])*1e3:.1f} ms), {n} findings, wave windows {sc.stats().wave_windows}")
    sc.free(d); sc.close()
