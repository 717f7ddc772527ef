"""The device-side exact replay (stringsext_amd/csrc/sx_replay_core.hpp), compiled as plain
host C++ by a test-only harness, must agree region by region with the product's host replayer
(sx_replay_shard_runs restricted to one region): same end position, same findings."""
import ctypes as C
import importlib.util
import os
import zlib
import random
import subprocess

import pytest

import refconfig as rc
import stringsext_amd as sx
import sxo_binding as sxo
from test_host_logic import soup, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NATIVE = os.path.join(ROOT, "tests", "native")


class ReplayParams(C.Structure):
    _fields_ = [("data", C.c_char_p), ("len", C.c_uint64), ("runs", C.POINTER(sx.Run)), ("n_runs", C.c_uint64),
                ("lo", C.c_uint64), ("hi", C.c_uint64), ("consumed0", C.c_uint64), ("stream0", C.c_uint64),
                ("slice_base", C.c_uint32), ("encoding", C.c_uint32), ("table", C.POINTER(C.c_uint16)),
                ("chars_min_nb", C.c_uint32), ("same_block", C.c_uint32), ("q", C.c_uint32), ("W", C.c_uint32),
                ("long_run", C.c_uint32), ("skip", C.c_uint32), ("grep_char", C.c_int32), ("mission_id", C.c_int32), ("file_id", C.c_int32),
                ("af_lo", C.c_uint64), ("af_hi", C.c_uint64), ("ubf", C.c_uint64),
                ("slot_of", C.c_void_p), ("n_heads", C.c_void_p), ("cache_arena", C.c_void_p), ("arena_bytes", C.c_uint64),
                ("head_list", C.c_void_p), ("max_windows", C.c_uint32), ("str_off_base", C.c_uint32), ("entry_skip", C.c_uint32),
                ("grid_flags", C.c_void_p), ("grid_sub", C.c_uint32),
                ("n_look", C.c_uint64), ("hard_list", C.c_void_p), ("n_hard", C.c_void_p)]   # (csrc/sx_device.hpp ReplayParams, every field: the core reads the whole struct)


def test_the_ctypes_mirror_of_replay_params_has_the_cores_size():
    """a field added to csrc/sx_device.hpp ReplayParams and not here shifts everything behind it (round 5: n_look read garbage)"""
    from native.build_harness import build_replay_core
    L = C.CDLL(build_replay_core())
    L.sxd_sizeof_replay_params.restype = C.c_uint64
    assert L.sxd_sizeof_replay_params() == C.sizeof(ReplayParams)


class RegionOut(C.Structure):
    _fields_ = [("end", C.c_uint64), ("n_find", C.c_uint32), ("n_bytes", C.c_uint32), ("status", C.c_uint32),
                ("pad", C.c_uint32)]


@pytest.fixture(scope="module")
def core():
    return load_core()


def load_core():
    from native.build_harness import build_replay_core
    so = build_replay_core()
    L = C.CDLL(so)
    L.sxd_replay_region_host.argtypes = [C.POINTER(ReplayParams), C.c_uint64, C.POINTER(RegionOut), C.POINTER(sx.Finding),
                                         C.POINTER(C.c_uint8), C.c_uint32, C.c_uint32]
    L.sxd_split_runs_host.restype = C.c_uint64
    L.sxd_split_runs_host.argtypes = [C.POINTER(ReplayParams), C.POINTER(sx.Run), C.c_uint64]
    return L


def split_runs(core, m, data, runs, stream0=0):
    """the runs cut into pieces at the window starts they cross, as the device cuts them (sx_replay_core.hpp kPieceCont)"""
    if not runs:
        return []
    P, _, _ = make_params(m, data, runs, stream0=stream0)
    cap = len(runs) + sum((b - a) // 12 + 2 for a, b, _ in runs)
    out = (sx.Run * cap)()
    n = core.sxd_split_runs_host(C.byref(P), out, cap)
    assert n <= cap
    return [(out[i].start, out[i].end, out[i].chars) for i in range(n)]


def sb_table(enc_id):
    """the product's decoder table (single byte: 128 entries; Big5 / EUC-JP: the blob)"""
    t = sx.decoder_table(enc_id)
    return t[0] if t else None


def ws(p, W):
    s = p // 4096 * 4096
    return s + (p - s) // W * W


def tricky(rng, n):
    """Strings framed by the byte patterns the shortcuts must get right: truncated and overlong
    UTF-8, stray continuation bytes, unpaired surrogates in front of BMP units, valid-but-
    rejected chars inside a decoder call, runs at window and slice edges."""
    frames = [b"\xe2\x82", b"\xf0\x9f\x98", b"\x80\x80", b"\xc0\xaf", b"\xed\xa0\x80", b"\xf4\x90", b"\x00\xd8", b"\xd8\x00",
              b"\x3d\xd8", b"\xd8\x3d", b"\x00\xdc", b"\x01\x02\x03", b"\t\x0b", b"\xc2\x85", b"\xe2\x80\xa8", b"\xff", b"\xfe\xff", b""]
    words = ["hello world!", "Ünïcödé-ßtring", "доброе утро", "a", "ab", "0123456789" * 8, "שלום עולם", "x" * 63, "y" * 64, "z" * 65,
             "mixed Ω≈ç√∫ text", "😀😀 astral 𝔘𝔫𝔦"]
    out = bytearray()
    while len(out) < n:
        w = rng.choice(words)
        enc = rng.choice(["utf-8", "utf-8", "utf-16-le", "utf-16-be", "koi8-r", "cp1252"])
        try:
            b = w.encode(enc)
        except UnicodeEncodeError:
            b = w.encode("utf-8")
        out += rng.choice(frames) + b + rng.choice(frames)
        r = rng.random()
        if r < 0.3:
            out += rng.randbytes(rng.randrange(1, 40))
        elif r < 0.4:
            out += b"\x00" * (-len(out) % 128)          # next string starts a window
        elif r < 0.45:
            out += b"\xff" * ((-len(out) - 5) % 4096)   # next string straddles a slice edge
    return bytes(out[:n])


CONFIGS = [
    dict(encodings=["utf-8"], chars_min="10", unicode_block_filter="African"),
    dict(encodings=["utf-8"], chars_min="4"),
    dict(encodings=["koi8-r"], chars_min="4"),
    dict(encodings=["ascii"], chars_min="5", output_line_len="10"),
    dict(encodings=["utf-16le"], chars_min="3", unicode_block_filter="All"),
    dict(encodings=["utf-16be"], chars_min="6", output_line_len="30", unicode_block_filter="Common"),
    dict(encodings=["utf-8"], chars_min="3", output_line_len="6", grep_char="47", unicode_block_filter="All"),
    dict(encodings=["utf-8"], chars_min="4", same_unicode_block=True, unicode_block_filter="All"),
    dict(encodings=["windows-1252"], chars_min="8", unicode_block_filter="Latin"),
    dict(encodings=["utf-8"], chars_min="70", output_line_len="64"),
    dict(encodings=["utf-16be"], chars_min="4", output_line_len="30", same_unicode_block=True, unicode_block_filter="Asian",
         ascii_filter="0x7ffffffe000000007ffffffe00000000"),
    dict(encodings=["utf-16le"], chars_min="3", same_unicode_block=True, unicode_block_filter="All"),
    dict(encodings=["iso-8859-5"], chars_min="4", output_line_len="6", same_unicode_block=True, ascii_filter="All-Ctrl",
         unicode_block_filter="Common"),
    # 64 < q <= 255: the replay core's QBIG instantiations (round 4; windows of up to 510 bytes)
    dict(encodings=["utf-8"], chars_min="4", output_line_len="100", unicode_block_filter="All"),
    dict(encodings=["utf-16le"], chars_min="10", output_line_len="255"),
    dict(encodings=["koi8-r"], chars_min="70", output_line_len="200", unicode_block_filter="Cyrillic"),
]


@pytest.mark.parametrize("flags", CONFIGS, ids=lambda f: "-".join(f["encodings"]) + "-n" + f["chars_min"])
@pytest.mark.parametrize("parity", [0, 1])
@pytest.mark.parametrize("skip", [1, 0], ids=["shortcuts", "every-byte"])
def test_device_replay_core_equals_host_replayer(core, flags, parity, skip):
    rng = random.Random(zlib.crc32(repr(sorted(flags.items())).encode()) & 0xFFFF)
    m = rc.missions(**flags)[0]
    if m["output_line_char_nb_max"] > 255:
        pytest.skip("device replay covers q <= 255")
    W = 2 * m["output_line_char_nb_max"]
    long_run = max(1, min(m["chars_min_nb"], m["output_line_char_nb_max"]))
    table = sb_table(m["encoding"]) if m["encoding"] >= 16 else None
    checked = 0
    for data in (synth(rng, 60_000, 1 / 150), soup(rng, 30_001), synth(rng, 20_000, 1 / 2000),
                 bytes(rng.choice(b"abcdefgh \x00") for _ in range(9_000)), tricky(rng, 40_000)):
        stream0 = parity  # odd stream offset of buffer byte 0 shifts the UTF-16 unit grid
        runs = sxo.runs(m, data, stream_parity=stream0 & 1, min_chars=long_run)
        arr = (sx.Run * max(1, len(runs)))(*[sx.Run(*t) for t in runs])
        P = ReplayParams(data, len(data), arr, len(runs), 0, len(data), m["counter_offset"] + stream0, stream0, 0,
                         m["encoding"], table, m["chars_min_nb"], int(m["require_same_unicode_block"]),
                         m["output_line_char_nb_max"], W, long_run, skip, -1 if m["grep_char"] is None else m["grep_char"],
                         0, 1, m["af"] & (2**64 - 1), m["af"] >> 64, m["ubf"], None, None, None, 0, None, 64)
        sc = sx.Scanner([m], device=sx.SX_HOST_ONLY)
        fbuf = (sx.Finding * 4096)()
        abuf = (C.c_uint8 * (1 << 20))()
        for i, r in enumerate(runs):
            want = ws(r[0], W)
            if want == 0 or (i > 0 and want <= ws(runs[i - 1][1] - 1, W)):
                continue  # first window: host only; chained: handled by the region before it
            o = RegionOut()
            rc_ = core.sxd_replay_region_host(C.byref(P), i, C.byref(o), fbuf, abuf, 4096, 1 << 20)
            assert rc_ == 0
            res, ends = sc.scan_shard(data, 0, 0, want + 1, start_at=[want], file_stream_off=stream0, file_id=1,
                                      runs_per_mission=[runs])
            host = [(f["position"], f["precision"], f["s"], f["completes"], f["slice_index"]) for f in res.findings()]
            res.free()
            arena = bytes(abuf[:o.n_bytes])
            dev = [(fbuf[k].position, sx.PRECISION[fbuf[k].precision],
                    arena[fbuf[k].str_off:fbuf[k].str_off + fbuf[k].str_len].decode("utf-8"),
                    bool(fbuf[k].completes_previous), fbuf[k].slice_index) for k in range(o.n_find)]
            if o.status == 3:
                continue  # too long for the device: given back to the host by design
            assert dev == host, (i, r, want)
            # the host reports len when no further region exists; otherwise the stop position must agree
            assert o.end == ends[0] or ends[0] == len(data), (i, r, want, o.end, ends[0])
            checked += 1
    assert checked > 0


def make_params(m, data, runs, stream0=0, skip=1):
    W = 2 * m["output_line_char_nb_max"]
    long_run = max(1, min(m["chars_min_nb"], m["output_line_char_nb_max"]))
    table = sb_table(m["encoding"]) if m["encoding"] >= 16 else None
    arr = (sx.Run * max(1, len(runs)))(*[sx.Run(*t) for t in runs])
    P = ReplayParams(data, len(data), arr, len(runs), 0, len(data), m["counter_offset"] + stream0, stream0, 0,
                     m["encoding"], table, m["chars_min_nb"], int(m["require_same_unicode_block"]),
                     m["output_line_char_nb_max"], W, long_run, skip, -1 if m["grep_char"] is None else m["grep_char"],
                     m["mission_id"], 1, m["af"] & (2**64 - 1), m["af"] >> 64, m["ubf"], None, None, None, 0, None, 64, 0)
    P._keep = (arr, table)
    return P, W, long_run


def emulate_device_stage_b(core, m, data, runs, skip=1, pieces=True, slabs=1):
    """What the device does with one Mission's runs, on the CPU: every head run (sx_replay_dev.hip
    region_is_chained) replays its region from derived state with the device core, then the sequential
    stitch rule (a region stands iff it begins at or behind the end of the last standing one).  The first
    region derives from the stream start, which is the true initial state.  Returns None if a region is
    given back to the host (too long).
    slabs > 1: as sx_stage_b.cpp device_replay_mission does it — the list is cut where a region begins
    (slab_cuts_kernel), every slab's kernels see only its runs (lookups may go on to the list's end: n_look),
    slab j owns the regions that begin in [where slab j-1 stopped, the window start of slab j+1's first run)."""
    if pieces:
        runs = split_runs(core, m, data, runs)
    P0, W, _ = make_params(m, data, runs, skip=skip)
    n = len(runs)
    chained = lambda i: i > 0 and (ws(runs[i][0], W) < runs[i - 1][1] if m["grep_char"] is None
                                   else ws(runs[i][0], W) <= ws(runs[i - 1][1] - 1, W) + W)
    cuts = [0]
    for j in range(1, slabs):
        i = n // slabs * j
        while i < n and chained(i):
            i += 1
        if cuts[-1] < i < n:
            cuts.append(i)
    cuts.append(n)
    fbuf = (sx.Finding * 8192)()
    abuf = (C.c_uint8 * (1 << 21))()
    out, E = [], 0
    for c0, c1 in zip(cuts, cuts[1:]):
        hi = ws(runs[c1][0], W) if c1 < n else len(data)
        P = ReplayParams.from_buffer_copy(P0)
        P.runs = C.cast(C.byref(P0._keep[0], c0 * C.sizeof(sx.Run)), C.POINTER(sx.Run))
        P.n_runs, P.n_look, P.lo, P.hi = c1 - c0, n - c0, E, hi
        for i in range(c0, c1):
            want = ws(runs[i][0], W)
            if i > c0 and chained(i):
                continue  # chained (sx_replay_core.hpp run_is_chained; the slice-end case of the -g form is covered by the stitch)
            if want < E or want >= hi:
                continue  # void: an earlier region ran over its start / the next slab's
            o = RegionOut()
            assert core.sxd_replay_region_host(C.byref(P), i - c0, C.byref(o), fbuf, abuf, 8192, 1 << 21) == 0
            if o.status == 3:
                return None
            arena = bytes(abuf[:o.n_bytes])
            out += [(fbuf[k].position, sx.PRECISION[fbuf[k].precision],
                     arena[fbuf[k].str_off:fbuf[k].str_off + fbuf[k].str_len].decode("utf-8"),
                     bool(fbuf[k].completes_previous), fbuf[k].slice_index) for k in range(o.n_find)]
            E = o.end
        E = max(E, min(hi, len(data)))
    return out


EMU_CONFIGS = CONFIGS + [
    dict(encodings=["ascii"], chars_min="1", ascii_filter="All-Ctrl+Wsp", unicode_block_filter="Common", grep_char="32",
         counter_offset="1000"),  # -g: a q-long stretch without the grep char ends the call's iteration (found by tools/gpu_fuzz.py)
    dict(encodings=["utf-8"], chars_min="1", output_line_len="8", grep_char="101"),
    dict(encodings=["ascii"], chars_min="4"),
    dict(encodings=["ascii"], chars_min="2", output_line_len="6", grep_char="32"),
    dict(encodings=["utf-8"], chars_min="1", output_line_len="8", unicode_block_filter="All"),
    dict(encodings=["utf-16le"], chars_min="1", unicode_block_filter="African", grep_char="0x65"),
]


@pytest.mark.parametrize("flags", EMU_CONFIGS, ids=lambda f: "-".join(f["encodings"]) + "-n" + f["chars_min"])
def test_device_pipeline_emulated_on_cpu_equals_the_oracle(core, flags):
    """Heads, derived entry states, region ends and the stitch together must give exactly the oracle's
    findings — this is where the premises of the region rules are checked (a region may end at a window
    start where nothing is pending although a run begins there: the next region derives that very state)."""
    from test_sharded_gloo import oracle_findings
    from test_gpu_parity import dense
    rng = random.Random(zlib.crc32(repr(sorted(flags.items())).encode()) & 0xFFFF)
    m = rc.missions(**flags)[0]
    if m["output_line_char_nb_max"] > 255:
        pytest.skip("device replay covers q <= 255")
    long_run = max(1, min(m["chars_min_nb"], m["output_line_char_nb_max"]))
    done = 0
    for data in (synth(rng, 40_000, 1 / 150), soup(rng, 20_001), synth(rng, 30_000, 1 / 40), tricky(rng, 30_000),
                 dense(rng, 30_000, 20, "abcdefgh XYZ019_-éжЖдяבשλ€😀"), bytes(rng.randrange(256) for _ in range(30_000))):
        runs = sxo.runs(m, data, stream_parity=0, min_chars=long_run)
        got = emulate_device_stage_b(core, m, data, runs)
        if got is None:
            continue
        want = [(p, pr, s, c, si) for p, pr, s, c, _, si in oracle_findings([m], data)]
        assert got == want, next(((a, b) for a, b in zip(got, want) if a != b), (len(got), len(want)))
        for slabs in (2, 5, 16):   # the same replayed in slabs (sx_stage_b.cpp device_replay_mission)
            got = emulate_device_stage_b(core, m, data, runs, slabs=slabs)
            assert got == want, (slabs, next(((a, b) for a, b in zip(got, want) if a != b), (len(got), len(want))))
        done += 1
    assert done > 0


def test_host_replay_understands_pieces_of_long_runs():
    """The run lists the device joins reach the host cut into pieces (kPieceCont): its sequential replay must not end a
    region at a piece boundary, and where it STARTS at one (a speculative part, a region the device gave back, the
    exit state) it derives the state from the run.  Text whose lines cross most window starts, several host parts."""
    from product_harness import run_cli_product
    rng = random.Random(99)
    words = [bytes(rng.choice(b"abcdefghijklmnopqrstuvwxyzABCDEFGH0123456789_-./:=") for _ in range(rng.randrange(2, 12))) for _ in range(300)]
    uni = ["Ünïcödé", "доброе утро", "שלום", "λόγος", "€uro"]
    out = bytearray()
    while len(out) < 9 * (1 << 20) + 12345:
        n = rng.choice([10, 30, 70, 127, 128, 129, 200, 400, 5000]); l = bytearray()
        while len(l) < n:
            l += (rng.choice(uni).encode() if rng.random() < 0.1 else rng.choice(words)) + b" "
        out += l[:n] + rng.choice([b"\n", b"\r\n", b"\x00", b"\xff", b"\n\n"])
    data = bytes(out)
    for flags in (dict(encodings=["utf-8"], chars_min="10"), dict(encodings=["ascii", "utf-8"], chars_min="4", output_line_len="20"),
                  dict(encodings=["utf-8"], chars_min="3", unicode_block_filter="All", output_line_len="7")):
        ms = rc.missions(**flags)
        want = sxo.run_cli(ms, [data], radix="x")
        for threads in (1, 4):
            assert run_cli_product(ms, [data], radix="x", pieces=True, replay_threads=threads) == want, (flags, threads)
        assert run_cli_product(ms, [data], radix="x", pieces=True, chunk_bytes=1 << 20) == want, flags
