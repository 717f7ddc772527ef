"""The wave-cooperative stage B (stringsext_amd/csrc/sx_wave_core.hpp: FindingCollection::from as bit arithmetic over
one window's masks), compiled as plain host C++ and driven exactly as sx_wave_dev.hip drives it (tests/native/
wave_core_host.cpp: wavefronts, batches of 64 windows, warm-up windows, entry states exchanged until consistent, count pass
then write pass), must give the oracle's findings — position, precision, `+`, string, slice — on every kind of input."""
import ctypes as C
import os
import random
import subprocess

import pytest

import refconfig as rc
import stringsext_amd as sx
import sxo_binding as sxo
from test_host_logic import soup, synth
from test_sharded_gloo import oracle_findings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NATIVE = os.path.join(ROOT, "tests", "native")


def load_wave():
    from native.build_harness import build_wave_core
    so = build_wave_core()
    L = C.CDLL(so)
    L.sxw_emulate.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32,
                              C.c_uint32, C.c_char_p, C.POINTER(C.c_uint16), C.c_int, C.c_int, C.POINTER(sx.Finding), C.c_uint64,
                              C.POINTER(C.c_uint8), C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32),
                              C.POINTER(C.c_uint64), C.c_int, C.POINTER(C.c_uint32), C.c_uint32, C.POINTER(C.c_uint32), C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_int, C.c_int, C.c_uint64]
    L.sxw_pack_state.restype = C.c_uint32
    L.sxw_pack_state.argtypes = [C.c_uint32] * 4
    return L


@pytest.fixture(scope="module")
def wave():
    return load_wave()


def wave_classes(m):
    """the product's class table for the Mission (sx_wave_classes), or None if the wave path does not cover it"""
    L = sx.lib()
    L.sx_wave_classes.argtypes, L.sx_wave_classes.restype = [C.POINTER(sx.Mission), C.POINTER(C.c_uint8)], C.c_int
    cm = sx.Mission.from_dict(dict(m, mission_id=m.get("mission_id", 0)))
    out = (C.c_uint8 * 512)()
    r = L.sx_wave_classes(C.byref(cm), out)
    assert r >= 0
    return (bytes(out) if r == 3 else bytes(out)[:256], r - 1) if r >= 1 else None   # (class table, family: 0 single byte, 1 UTF-8, 4 the two-byte family)


def wave_swar(m):
    """the Mission's classes as SWAR ranges (csrc/sx_device.hpp WvSwar, 26 words) if the product classifies it that way, else None"""
    L = sx.lib()
    L.sx_wave_swar.argtypes, L.sx_wave_swar.restype = [C.POINTER(sx.Mission), C.POINTER(C.c_uint32)], C.c_int
    out = (C.c_uint32 * 26)()
    cm = sx.Mission.from_dict(dict(m, mission_id=m.get("mission_id", 0)))
    r = L.sx_wave_swar(C.byref(cm), out)
    assert r >= 0
    return out if r == 1 else None


def wave_pairs2(m):
    L = sx.lib()
    L.sx_wave_pair_codes2.argtypes, L.sx_wave_pair_codes2.restype = [C.POINTER(sx.Mission), C.POINTER(C.c_uint32)], C.c_void_p
    out = (C.c_uint32 * 4096)()
    cm = sx.Mission.from_dict(dict(m, mission_id=m.get("mission_id", 0)))
    return out if L.sx_wave_pair_codes2(C.byref(cm), out) else None


def wave_pairs(m):
    L = sx.lib()
    L.sx_wave_pair_codes.argtypes, L.sx_wave_pair_codes.restype = [C.POINTER(sx.Mission), C.POINTER(C.c_uint32)], C.c_void_p
    out = (C.c_uint32 * 8192)()
    cm = sx.Mission.from_dict(dict(m, mission_id=m.get("mission_id", 0)))
    return out if L.sx_wave_pair_codes(C.byref(cm), out) else None


PREC = {0: "Before", 1: "Exact", 2: "After"}


def emulate(L, m, data, nwin=508, skip_idle=1, g_lo=0, inject=0, consumed0=None, swar=True, may_give_up=False):
    lut, family = wave_classes(m)
    t = sx.decoder_table(m["encoding"])
    table = t[0] if t else None
    q = m["output_line_char_nb_max"]
    cap_f = len(data) + 64
    cap_a = 4 * len(data) + 4096
    fout = (sx.Finding * cap_f)()
    aout = (C.c_uint8 * cap_a)()
    nf, nb, bad = C.c_uint64(), C.c_uint64(), C.c_uint64()
    fin, rounds = C.c_uint32(), C.c_uint32()
    rcode = L.sxw_emulate(data, len(data), m["counter_offset"] if consumed0 is None else consumed0, 0, 2 * q, q, m["chars_min_nb"], g_lo,
                          inject, nwin, lut, table, 0, 1, fout, cap_f, aout, cap_a, C.byref(nf), C.byref(nb), C.byref(fin), C.byref(bad),
                          skip_idle, C.byref(rounds), family, wave_pairs(m) if family == 4 else None, m["encoding"], 0,
                          wave_swar(m) if swar or family == 5 else None, wave_pairs2(m) if (swar and family == 4) or family == 5 else None,
                          -1 if m.get("grep_char") is None else m["grep_char"],
                          1 if m.get("require_same_unicode_block") and (m.get("grep_char") is None or bin(m["ubf"] & 0x001FFFFFFFFFFFFC).count("1") <= 31) else 0,
                          m["ubf"])   # (-g AND -r: the kernels apply -r only with at most 31 lead bytes — five bits of state; a Mission the product still sends here has one lead byte at most)
    if may_give_up and rcode in (-9, -10):   # the wavefronts gave the buffer back (UTF-16: a case the masks cannot say; -10: the two-byte family after repairs, no descriptors)
        return None, dict(gave_up=True)
    assert rcode == 0, rcode
    arena = bytes(aout[:nb.value])
    got = []
    for i in range(nf.value):
        f = fout[i]
        got.append((f.position, PREC[f.precision], arena[f.str_off:f.str_off + f.str_len].decode("utf-8"), bool(f.completes_previous), 0,
                    f.slice_index))
    return got, dict(final=fin.value, bad=bad.value, rounds=rounds.value)


def text_lines(rng, n, lo=10, hi=120, alphabet=b"abcdefghijklmnopqrstuvwxyzABCDEFGH0123456789_-./:= "):
    out = bytearray()
    while len(out) < n:
        out += bytes(rng.choice(alphabet) for _ in range(rng.randrange(lo, hi))) + b"\n"
    return bytes(out[:n])


def records(rng, n, rec, fill):
    """fixed-width records on the window grid: `rec` accepted bytes, then rejected filler up to the next multiple of 64"""
    out = bytearray()
    while len(out) < n:
        out += b"x" * rec
        out += bytes([fill]) * (-len(out) % 64 or 64)
    return bytes(out[:n])


def inputs(rng):
    yield "random", rng.randbytes(70_000)
    yield "text", text_lines(rng, 90_000)
    yield "long lines", text_lines(rng, 60_000, 100, 700)
    yield "soup", soup(rng, 50_000)
    yield "synth dense", synth(rng, 80_000, 1 / 60)
    yield "all accepted", b"A" * 20_000
    yield "records 64", records(rng, 40_000, 64, 0)
    yield "records 60", records(rng, 40_000, 60, 0xFF)
    yield "records 128", records(rng, 40_000, 128, 1)
    yield "high bytes", bytes(rng.choice([0x41, 0x42, 0xC0, 0xE1, 0xFF, 0x98, 0x0A, 0x20]) for _ in range(50_000))
    yield "short tail", text_lines(rng, 4096 * 3 + 77)
    words = ["hello world!", "Ünïcödé-ßtring", "доброе утро", "שלום עולם", "中文字符串测试", "😀😀 astral 𝔘𝔫𝔦", "Բարեւ աշխարհ", "mixed Ω≈ç√∫ text", "x" * 70]
    yield "utf-8 text", ("".join(rng.choice(words) + rng.choice([" ", "\n", "\t", " — "]) for _ in range(6000))).encode()
    yield "utf-8 4-byte", ("😀" * 50 + "\n" + "𝔘𝔫𝔦" * 40 + "a").encode() * 40
    frames = [b"\xe2\x82", b"\xf0\x9f\x98", b"\x80\x80", b"\xc0\xaf", b"\xed\xa0\x80", b"\xf4\x90", b"\xff", b"\xe0\x80", b"\xc3", b"\xf0\x90\x80", b""]
    yield "utf-8 broken", b"".join(rng.choice(frames) + rng.choice(words).encode() + rng.choice(frames) + rng.randbytes(rng.randrange(0, 6)) for _ in range(5000))
    # a slice ends with [leftover][a lead byte still pending] and the next one starts with a byte that sequence rejects: an empty
    # first call, then a second one at byte 0 whose precision the probe decides against [leftover][output] (finding_collection.rs:176-207)
    d = bytearray(text_lines(rng, 4096 * 24))
    for k in range(1, 24):
        left = rng.choice([b"7", "Ä".encode(), "ÄÄ".encode(), b"ab", "é".encode(), b""])
        d[4096 * k - len(left) - 2:4096 * k] = b"\x00" + left + b"\xc3"
        d[4096 * k:4096 * k + 40] = rng.choice(["Ä" * 20, "ÄÖ" * 10, "éÄ" * 10, "Äa" * 13 + "Ä"]).encode()
    yield "slice-start probe", bytes(d)
    yield "tiny", b"hello world, this is tiny\n"
    yield "one byte", b"a"


MISSIONS = [
    dict(encodings=["ascii"], chars_min="4"),
    dict(encodings=["ascii"], chars_min="1"),
    dict(encodings=["ascii"], chars_min="10", output_line_len="10"),
    dict(encodings=["ascii"], chars_min="3", output_line_len="6"),
    dict(encodings=["ascii"], chars_min="5", output_line_len="30", ascii_filter="All"),
    dict(encodings=["x-user-defined"], chars_min="4", unicode_block_filter="All"),
    dict(encodings=["koi8-r"], chars_min="10", unicode_block_filter="Cyrillic"),
    dict(encodings=["koi8-r"], chars_min="4"),
    dict(encodings=["windows-1252"], chars_min="4", unicode_block_filter="Latin"),
    dict(encodings=["windows-1253"], chars_min="2", output_line_len="8", unicode_block_filter="All"),
    dict(encodings=["windows-874"], chars_min="6", unicode_block_filter="All"),
    dict(encodings=["iso-8859-7"], chars_min="64", output_line_len="64", unicode_block_filter="All"),
    dict(encodings=["ibm866"], chars_min="7", output_line_len="32", ascii_filter="None", unicode_block_filter="All"),
    dict(encodings=["utf-8"], chars_min="10"),
    dict(encodings=["utf-8"], chars_min="4", unicode_block_filter="All"),
    dict(encodings=["utf-8"], chars_min="3", output_line_len="6", unicode_block_filter="All"),
    dict(encodings=["utf-8"], chars_min="10", unicode_block_filter="African"),
    dict(encodings=["utf-8"], chars_min="2", output_line_len="16", unicode_block_filter="Cjk", ascii_filter="None"),
    dict(encodings=["utf-8"], chars_min="20", output_line_len="20", unicode_block_filter="Uncommon"),
    # -r where it cannot break a string: at most one UTF-8 lead byte passes the filter (x-user-defined: every character >= 0x80 begins with EF)
    dict(encodings=["ascii"], chars_min="4", same_unicode_block=True),
    dict(encodings=["x-user-defined"], chars_min="3", output_line_len="20", unicode_block_filter="All", same_unicode_block=True),
    dict(encodings=["utf-8"], chars_min="3", unicode_block_filter="0x10000", same_unicode_block=True),   # (lead byte D0 only)
]


# -g (round 5): a string counts only if it holds the grep char; a line of q chars without it that neither completes the string before nor
# is carried on ends SplitStr's walk over the call's text (helper.rs:410-415).  The first two are the reference's functional tests 1 and 2
# (tests/functional/run-tests:11-29), one Mission each.
GREP_MISSIONS = [
    dict(encodings=["utf-8"], output_line_len="16", grep_char="63", ascii_filter="All-Ctrl", unicode_block_filter="Common"),
    dict(encodings=["utf-8"], chars_min="10", output_line_len="32", grep_char="58", ascii_filter="All-Ctrl", unicode_block_filter="Common"),
    dict(encodings=["ascii"], chars_min="4", grep_char="47"),
    dict(encodings=["ascii"], chars_min="3", output_line_len="6", grep_char="101"),
    dict(encodings=["ascii"], chars_min="6", output_line_len="6", grep_char="32"),
    dict(encodings=["ascii"], chars_min="4", grep_char="10"),                                   # a grep char the filter rejects: it counts for the string it ends
    dict(encodings=["koi8-r"], chars_min="5", unicode_block_filter="Cyrillic", grep_char="32"),
    dict(encodings=["windows-1253"], chars_min="2", output_line_len="8", unicode_block_filter="All", grep_char="97"),
    dict(encodings=["utf-8"], chars_min="4", unicode_block_filter="All", grep_char="32"),
    dict(encodings=["utf-8"], chars_min="3", output_line_len="6", unicode_block_filter="All", grep_char="111"),
    dict(encodings=["utf-8"], chars_min="2", output_line_len="16", unicode_block_filter="Cjk", ascii_filter="None", grep_char="32"),   # ... rejected too
]
GREP_UTF16 = [
    dict(encodings=["utf-16le"], output_line_len="16", grep_char="63", ascii_filter="All-Ctrl", unicode_block_filter="Common"),
    dict(encodings=["utf-16be"], chars_min="10", output_line_len="32", grep_char="58", ascii_filter="All-Ctrl", unicode_block_filter="Common"),
    dict(encodings=["utf-16be"], chars_min="2", unicode_block_filter="All", output_line_len="10", grep_char="97"),
    dict(encodings=["utf-16le"], chars_min="3", unicode_block_filter="Cjk", ascii_filter="None", grep_char="32"),
]
GREP_DBCS = [
    ("big5", dict(encodings=["big5"], chars_min="3", output_line_len="8", unicode_block_filter="Asian", grep_char="32")),
    ("shift_jis", dict(encodings=["shift_jis"], chars_min="4", unicode_block_filter="All", grep_char="65")),
    ("euc-kr", dict(encodings=["euc-kr"], chars_min="4", unicode_block_filter="All", grep_char="32")),
    ("euc-jp", dict(encodings=["euc-jp"], chars_min="3", output_line_len="8", unicode_block_filter="Cjk", grep_char="65")),
]


def grep_text(rng, n, g):
    """lines of every length with and without the grep char, next to window edges"""
    out = bytearray()
    alpha = bytes(c for c in b"abcdefghijklmnopqrstuvwxyzABCDEFGH0123456789_-.=" if c != g)
    while len(out) < n:
        ln = bytearray(rng.choice(alpha) for _ in range(rng.choice([2, 5, 6, 7, 12, 16, 17, 31, 32, 33, 63, 64, 65, 100, 130, 200])))
        for _ in range(rng.choice([0, 0, 1, 1, 2, 5])):
            ln[rng.randrange(len(ln))] = g
        out += ln + rng.choice([b"\n", b"\x00", b"\xff", bytes([g]), b"\n\n", b"\x01\x02"])
    return bytes(out[:n])


@pytest.mark.parametrize("gi", range(len(GREP_MISSIONS)))
def test_emulated_wave_pipeline_with_a_grep_char(wave, gi):
    m = rc.missions(**GREP_MISSIONS[gi])[0]
    assert wave_classes(m) is not None
    rng = random.Random(9000 + gi)
    golden = open(os.path.join(ROOT, "tests", "golden", "input1"), "rb").read()
    extra = [("grep text", grep_text(rng, 80_000, m["grep_char"])), ("input1", golden),
             ("all grep", bytes([m["grep_char"]]) * 9000), ("no grep", bytes(c for c in text_lines(rng, 30_000) if c != m["grep_char"]))]
    for name, data in list(inputs(rng)) + extra:
        want = oracle_findings([dict(m, mission_id=0)], data)
        for nwin, skip in ((508, 1), (60, 0), (7, 1)):
            got, info = emulate(wave, m, data, nwin=nwin, skip_idle=skip, swar=nwin != 60)
            assert info["bad"] == 0, (name, nwin, info)
            assert got == want, (name, nwin, skip, next(((a, b) for a, b in zip(got, want) if a != b), (len(got), len(want))))


@pytest.mark.parametrize("gi", range(len(GREP_UTF16)))
def test_emulated_wave_pipeline_utf16_with_a_grep_char(wave, gi):
    m = rc.missions(**GREP_UTF16[gi])[0]
    be = GREP_UTF16[gi]["encodings"][0].endswith("be")
    codec = "utf-16-be" if be else "utf-16-le"
    rng = random.Random(9500 + gi)
    golden2 = open(os.path.join(ROOT, "tests", "golden", "input2"), "rb").read()
    datas = [("grep text", grep_text(rng, 30_000, m["grep_char"]).decode("latin-1").encode(codec)), ("soup", utf16_soup(rng, 30_000, be)),
             ("text", text_lines(rng, 30_000).decode("latin-1").encode(codec)), ("input2", golden2[:len(golden2) // 2 * 2]),
             ("astral", ("a\U0001F600b:\U00020000\U0001F601c?d \u4e2d" * 2000).encode(codec)), ("random", rng.randbytes(60_000))]
    for name, data in datas:
        want = oracle_findings([dict(m, mission_id=0)], data)
        for nwin, skip in ((508, 1), (60, 0), (7, 1)):
            got, info = emulate(wave, m, data, nwin=nwin, skip_idle=skip, may_give_up=True)
            if got is None:
                assert name in ("soup", "random", "input2"), name
                continue
            assert info["bad"] == 0, (name, nwin, info)
            assert got == want, (name, nwin, skip, next(((a, b) for a, b in zip(got, want) if a != b), (len(got), len(want))))


# -r (round 5, helper.rs:279-296): a multi-byte character that passes the filter but whose lead byte differs from the one of the multi-byte
# character before it in the same walk ends the string in front of it and begins the next one
SAME_MISSIONS = [
    dict(encodings=["utf-8"], chars_min="4", unicode_block_filter="Cyrillic", same_unicode_block=True),
    dict(encodings=["utf-8"], chars_min="2", output_line_len="8", unicode_block_filter="All", same_unicode_block=True),
    dict(encodings=["utf-8"], chars_min="3", output_line_len="16", unicode_block_filter="All", ascii_filter="None", same_unicode_block=True),
    dict(encodings=["utf-8"], chars_min="1", output_line_len="6", unicode_block_filter="Common", same_unicode_block=True),
    dict(encodings=["utf-8"], chars_min="5", unicode_block_filter="All", ascii_filter="All-Ctrl", same_unicode_block=True),
    dict(encodings=["koi8-r"], chars_min="4", unicode_block_filter="Cyrillic", same_unicode_block=True),
    dict(encodings=["koi8-r"], chars_min="2", output_line_len="6", unicode_block_filter="All", same_unicode_block=True),
    dict(encodings=["windows-1253"], chars_min="2", output_line_len="8", unicode_block_filter="All", same_unicode_block=True),
    dict(encodings=["windows-1252"], chars_min="3", unicode_block_filter="All", ascii_filter="None", same_unicode_block=True),
]
SAME_UTF16 = [
    dict(encodings=["utf-16le"], chars_min="3", unicode_block_filter="All", same_unicode_block=True),
    dict(encodings=["utf-16be"], chars_min="4", output_line_len="10", unicode_block_filter="Cyrillic", same_unicode_block=True),
    dict(encodings=["utf-16le"], chars_min="2", output_line_len="6", unicode_block_filter="All", ascii_filter="None", same_unicode_block=True),
    dict(encodings=["utf-16be"], chars_min="1", output_line_len="8", unicode_block_filter="Common", same_unicode_block=True),
]
# ... and with -g as well (round 5, last): a sub-stretch counts only with the grep char (filters with at most 31 multi-byte lead bytes: the
# state has five bits for the lead code then; `-u All -g -r` keeps the check per buffer)
SAME_GREP_MISSIONS = [
    dict(encodings=["utf-8"], chars_min="4", unicode_block_filter="Cyrillic", same_unicode_block=True, grep_char="32"),
    dict(encodings=["utf-8"], chars_min="2", output_line_len="8", unicode_block_filter="Common", same_unicode_block=True, grep_char="101"),
    dict(encodings=["utf-8"], chars_min="3", output_line_len="16", unicode_block_filter="Common", ascii_filter="None", same_unicode_block=True, grep_char="32"),   # a grep char the filter rejects
    dict(encodings=["koi8-r"], chars_min="3", output_line_len="10", unicode_block_filter="Common", same_unicode_block=True, grep_char="32"),
    dict(encodings=["windows-1253"], chars_min="2", output_line_len="8", unicode_block_filter="Common", same_unicode_block=True, grep_char="97"),
    dict(encodings=["utf-16le"], chars_min="3", unicode_block_filter="Common", same_unicode_block=True, grep_char="32"),
    dict(encodings=["utf-16be"], chars_min="2", output_line_len="6", unicode_block_filter="Common", ascii_filter="None", same_unicode_block=True, grep_char="48"),
]
SCRIPTS = ["abcdefghijklmnopqrstuvwxyz 0123456789", "\u0430\u0431\u0432\u0433\u0434\u0435\u0436\u0437\u0438\u0439\u043a\u043b\u043c\u043d\u043e\u043f",   # Cyrillic, lead D0
           "\u0440\u0441\u0442\u0443\u0444\u0445\u0446\u0447\u0448\u0449\u044a\u044b\u044c\u044d\u044e\u044f",                                       # ... D1
           "\u03b1\u03b2\u03b3\u03b4\u03b5\u03b6\u03b7\u03b8\u03b9\u03ba\u03bb\u03bc\u03bd\u03be\u03bf", "\u03c0\u03c1\u03c3\u03c4\u03c5\u03c6\u03c7\u03c8\u03c9",   # Greek CE / CF
           "\u00e0\u00e9\u00ee\u00f5\u00fc\u00df\u00c6", "\u00a1\u00a9\u00ae\u00b5\u00bf",                                                              # Latin-1 C3 / C2
           "\u4e2d\u6587\u5b57\u7b26", "\u3042\u3044\u3046\u30a2\u30a4", "\u20ac\u2013\u2022", "\U0001F600\U0001F601\U00020000", "\x01\x02\x7f", "\n"]


def same_text(rng, n_chars, weights=(30, 20, 20, 6, 6, 6, 3, 4, 3, 3, 2, 2, 4), runs=(1, 1, 2, 3, 4, 5, 8, 13, 40)):
    """runs of characters of one script each: lead bytes change inside lines, at their ends, next to window edges"""
    out = []
    k = 0
    while k < n_chars:
        sc = rng.choices(SCRIPTS, weights)[0]
        ln = rng.choice(runs)
        out.append("".join(rng.choice(sc) for _ in range(ln)))
        k += ln
    return "".join(out)


def russian(rng, n_chars):
    words = ["\u043f\u0440\u0438\u0432\u0435\u0442", "\u043c\u0438\u0440", "\u0441\u0442\u0440\u043e\u043a\u0430", "\u0434\u0430", "\u043d\u0435\u0442", "\u0430\u0431\u0432\u0433\u0434", "\u0440\u0441\u0442\u0443\u0444", "hello", "x", "42"]
    out = []
    k = 0
    while k < n_chars:
        wd = rng.choice(words)
        out.append(wd + rng.choice([" ", " ", " ", ", ", ".\n", "\n"]))
        k += len(wd) + 1
    return "".join(out)


@pytest.mark.parametrize("si", range(len(SAME_MISSIONS)))
def test_emulated_wave_pipeline_with_same_unicode_block(wave, si):
    m = rc.missions(**SAME_MISSIONS[si])[0]
    assert wave_classes(m) is not None
    codec = SAME_MISSIONS[si]["encodings"][0]
    rng = random.Random(9700 + si)
    golden = open(os.path.join(ROOT, "tests", "golden", "input1"), "rb").read()
    enc = lambda t: t.encode(codec, errors="replace" if codec != "utf-8" else "strict")
    extra = [("scripts", enc(same_text(rng, 40_000))), ("scripts, long runs", enc(same_text(rng, 30_000, runs=(1, 7, 30, 64, 65, 130)))),
             ("russian", enc(russian(rng, 30_000))), ("input1", golden),
             ("two leads", enc("".join(rng.choice("\u043f\u0440") for _ in range(20_000)))),
             ("scripts, no ascii", enc(same_text(rng, 30_000, weights=(0, 20, 20, 6, 6, 6, 3, 4, 3, 3, 2, 1, 1))))]
    for name, data in list(inputs(rng)) + extra:
        want = oracle_findings([dict(m, mission_id=0)], data)
        for nwin, skip in ((508, 1), (60, 0), (7, 1)):
            got, info = emulate(wave, m, data, nwin=nwin, skip_idle=skip, swar=nwin != 60)
            assert info["bad"] == 0, (name, nwin, info)
            assert got == want, (name, nwin, skip, next(((a, b) for a, b in zip(got, want) if a != b), (len(got), len(want))))


@pytest.mark.parametrize("si", range(len(SAME_GREP_MISSIONS)))
def test_emulated_wave_pipeline_with_same_unicode_block_and_a_grep_char(wave, si):
    kw = SAME_GREP_MISSIONS[si]
    m = rc.missions(**kw)[0]
    assert wave_classes(m) is not None
    codec = kw["encodings"][0]
    u16 = codec.startswith("utf-16")
    rng = random.Random(9900 + si)
    enc = lambda t: t.encode(codec, errors="replace" if not codec.startswith("utf-") else "strict")
    g = chr(m["grep_char"])
    def sprinkle(t, every):   # the grep char every few characters, and stretches without it
        out = []
        for i in range(0, len(t), 400):
            piece = t[i:i + 400]
            out.append(piece if (i // 400) % 3 == 2 else "".join(c if rng.randrange(every) else g for c in piece))
        return "".join(out)
    datas = [("scripts", enc(sprinkle(same_text(rng, 40_000), 9))), ("long runs", enc(sprinkle(same_text(rng, 30_000, runs=(1, 7, 30, 64, 65, 130)), 25))),
             ("russian", enc(russian(rng, 30_000))), ("grep text", enc(grep_text(rng, 30_000, m["grep_char"]).decode("latin-1"))),
             ("random", rng.randbytes(40_000))]
    if not u16:
        datas += list(inputs(rng))
    for name, data in datas:
        want = oracle_findings([dict(m, mission_id=0)], data)
        for nwin, skip in ((508, 1), (60, 0), (7, 1)):
            got, info = emulate(wave, m, data, nwin=nwin, skip_idle=skip, swar=nwin != 60, may_give_up=u16)
            if got is None:
                assert name == "random", name
                continue
            assert info["bad"] == 0, (name, nwin, info)
            assert got == want, (name, nwin, skip, next(((a, b) for a, b in zip(got, want) if a != b), (len(got), len(want))))


@pytest.mark.parametrize("si", range(len(SAME_UTF16)))
def test_emulated_wave_pipeline_utf16_with_same_unicode_block(wave, si):
    m = rc.missions(**SAME_UTF16[si])[0]
    assert wave_classes(m) is not None
    be = SAME_UTF16[si]["encodings"][0].endswith("be")
    codec = "utf-16-be" if be else "utf-16-le"
    rng = random.Random(9800 + si)
    golden2 = open(os.path.join(ROOT, "tests", "golden", "input2"), "rb").read()
    datas = [("scripts", same_text(rng, 30_000).encode(codec)), ("scripts, long runs", same_text(rng, 20_000, runs=(1, 7, 30, 64, 65, 130)).encode(codec)),
             ("russian", russian(rng, 20_000).encode(codec)), ("soup", utf16_soup(rng, 30_000, be)), ("input2", golden2[:len(golden2) // 2 * 2]),
             ("astral", ("a\U0001F600b:\U00020000\U0001F601c?d \u4e2d" * 2000).encode(codec)), ("random", rng.randbytes(60_000))]
    for name, data in datas:
        want = oracle_findings([dict(m, mission_id=0)], data)
        for nwin, skip in ((508, 1), (60, 0), (7, 1)):
            got, info = emulate(wave, m, data, nwin=nwin, skip_idle=skip, may_give_up=True)
            if got is None:
                assert name in ("soup", "random", "input2"), name
                continue
            assert info["bad"] == 0, (name, nwin, info)
            assert got == want, (name, nwin, skip, next(((a, b) for a, b in zip(got, want) if a != b), (len(got), len(want))))



@pytest.mark.parametrize("gi", range(len(GREP_DBCS)))
def test_emulated_wave_pipeline_two_byte_family_with_a_grep_char(wave, gi):
    from test_dbcs import soup as dbcs_soup, TEXT, CODEC
    enc, flags = GREP_DBCS[gi]
    m = rc.missions(**flags)[0]
    rng = random.Random(9800 + gi)
    txt = TEXT[enc].encode(CODEC[enc], "ignore")
    datas = [("soup", dbcs_soup(enc, rng, 80_000)), ("random", rng.randbytes(60_000)), ("text", (txt + b"\n") * (40_000 // (len(txt) + 1))),
             ("ascii", grep_text(rng, 40_000, m["grep_char"]))]
    for name, data in datas:
        want = oracle_findings([dict(m, mission_id=0)], data)
        for nwin, skip in ((508, 1), (60, 0), (7, 1)):
            got, info = emulate(wave, m, data, nwin=nwin, skip_idle=skip, swar=nwin != 60)
            assert info["bad"] == 0, (name, nwin, info)
            assert got == want, (enc, name, nwin, skip, next(((a, b) for a, b in zip(got, want) if a != b), (len(got), len(want))))


DBCS_MISSIONS = [
    ("big5", dict(encodings=["big5"], chars_min="10", unicode_block_filter="Cjk")),
    ("big5", dict(encodings=["big5"], chars_min="3", output_line_len="8", unicode_block_filter="Asian")),
    ("big5", dict(encodings=["big5"], chars_min="2", output_line_len="6", unicode_block_filter="Cjk", ascii_filter="None")),
    ("shift_jis", dict(encodings=["shift_jis"], chars_min="4", unicode_block_filter="All")),
    ("shift_jis", dict(encodings=["shift_jis"], chars_min="10", unicode_block_filter="Kana")),
    ("shift_jis", dict(encodings=["shift_jis"], chars_min="2", output_line_len="6", unicode_block_filter="All", ascii_filter="All")),
    ("euc-kr", dict(encodings=["euc-kr"], chars_min="4", unicode_block_filter="All")),
    ("euc-kr", dict(encodings=["euc-kr"], chars_min="20", output_line_len="20", unicode_block_filter="Hangul")),
    ("euc-jp", dict(encodings=["euc-jp"], chars_min="10", unicode_block_filter="Asian")),
    ("euc-jp", dict(encodings=["euc-jp"], chars_min="3", output_line_len="8", unicode_block_filter="Cjk")),
    ("euc-jp", dict(encodings=["euc-jp"], chars_min="2", output_line_len="6", unicode_block_filter="Kana", ascii_filter="None")),
    ("euc-jp", dict(encodings=["euc-jp"], chars_min="4", unicode_block_filter="None")),
]


@pytest.mark.parametrize("di", range(len(DBCS_MISSIONS)))
def test_emulated_wave_pipeline_two_byte_family(wave, di):
    """Big5, Shift_JIS, EUC-KR: token starts composed lane to lane, pair codes, pending lead bytes at window and slice starts"""
    from test_dbcs import soup as dbcs_soup, TEXT, CODEC
    enc, flags = DBCS_MISSIONS[di]
    m = rc.missions(**flags)[0]
    assert wave_classes(m)[1] == (5 if enc == "euc-jp" else 4)
    rng = random.Random(3000 + di)
    txt = TEXT[enc].encode(CODEC[enc], "ignore")
    datas = [("soup", dbcs_soup(enc, rng, 120_000)), ("random", rng.randbytes(90_000)), ("text", (txt + b"\n") * (60_000 // (len(txt) + 1))),
             ("text no ascii", txt.replace(b" ", b"").replace(b"\n", b"") * 40), ("lead bytes", b"\xa4" * 5000 + b"A" + b"\xa4\xa4" * 3000 + b"\x00" * 300),
             ("three-byte tokens", (b"\x8f\xb0\xa1\x8f\xb0\xa2\x8e\xb1\x8f\xa1\x41\x8f\x41\xa4\xa2" * 9 + b"\n") * 700),
             ("ascii", text_lines(rng, 50_000)), ("short tail", dbcs_soup(enc, rng, 4096 * 3 + 77))]
    for name, data in datas:
        want = oracle_findings([dict(m, mission_id=0)], data)
        for nwin, skip in ((508, 1), (60, 0), (7, 1)):
            got, info = emulate(wave, m, data, nwin=nwin, skip_idle=skip, swar=nwin != 60)
            assert info["bad"] == 0, (name, nwin, info)
            assert got == want, (enc, name, nwin, skip, next(((a, b) for a, b in zip(got, want) if a != b), (len(got), len(want))))


def utf16_soup(rng, n_units, be, weights=(50, 20, 8, 8, 6, 8)):
    """units of every kind: ASCII, BMP characters of two and three UTF-8 bytes, lone high / low surrogates, pairs, anything"""
    import struct
    out = bytearray()
    for _ in range(n_units):
        k = rng.choices(range(6), weights)[0]
        if k == 0: us = [rng.choice(b"abcdefghij XYZ019\n")]
        elif k == 1: us = [rng.choice([0xE9, 0x416, 0x4E2D, 0x3042, 0x7FF, 0x800, 0xFFFD, 0x2028, 0x80])]
        elif k == 2: us = [rng.randrange(0xD800, 0xDC00)]
        elif k == 3: us = [rng.randrange(0xDC00, 0xE000)]
        elif k == 4: us = [rng.randrange(0xD800, 0xDC00), rng.randrange(0xDC00, 0xE000)]
        else: us = [rng.randrange(0, 0x10000)]
        for u in us:
            out += struct.pack(">H" if be else "<H", u)
    return bytes(out)


UTF16_MISSIONS = [
    dict(encodings=["utf-16le"], chars_min="4"),
    dict(encodings=["utf-16be"], chars_min="2", unicode_block_filter="All", output_line_len="10"),
    dict(encodings=["utf-16le"], chars_min="1", unicode_block_filter="All", output_line_len="33"),
    dict(encodings=["utf-16be"], chars_min="3", unicode_block_filter="Cjk", ascii_filter="None"),
    dict(encodings=["utf-16le"], chars_min="2", unicode_block_filter="0xFFFF000000000000", ascii_filter="None", output_line_len="6"),   # astral characters only
]


@pytest.mark.parametrize("ui", range(len(UTF16_MISSIONS)))
def test_emulated_wave_pipeline_utf16(wave, ui):
    """UTF-16LE / BE on the unit grid: surrogates alone, in pairs, across window ends; the slow mode behind a pending high surrogate and the
    character it keeps for the next call; every window's marks against the decoder's state machine; findings against the oracle"""
    m = rc.missions(**UTF16_MISSIONS[ui])[0]
    assert wave_classes(m)[1] == 2 and len(wave_classes(m)[0]) == 512
    be = UTF16_MISSIONS[ui]["encodings"][0].endswith("be")
    codec = "utf-16-be" if be else "utf-16-le"
    rng = random.Random(7000 + ui)
    datas = [("soup", utf16_soup(rng, 40_000, be)), ("random", rng.randbytes(90_000)),
             ("text", text_lines(rng, 40_000).decode("latin-1").encode(codec)),
             ("astral", ("a\U0001F600b\U00020000\U0001F601cd \u4e2d" * 3000).encode(codec)),
             ("high surrogates", utf16_soup(rng, 30_000, be, (40, 10, 14, 10, 10, 4))), ("short tail", utf16_soup(rng, 2048 * 3 + 39, be))]
    gave_up = 0
    for name, data in datas:
        want = oracle_findings([dict(m, mission_id=0)], data)
        for nwin, skip in ((508, 1), (60, 0), (7, 1)):
            got, info = emulate(wave, m, data, nwin=nwin, skip_idle=skip, may_give_up=True)
            if got is None:   # (a kept character at a window's end, seven high surrogates in a row: the product takes the other path)
                assert name in ("soup", "high surrogates", "random", "short tail"), name
                gave_up += 1
                continue
            assert info["bad"] == 0, (name, nwin, info)
            assert got == want, (name, nwin, skip, next(((a, b) for a, b in zip(got, want) if a != b), (len(got), len(want))))
    assert gave_up <= 9


def test_big5_missions_that_accept_the_two_code_point_tokens_stay_on_the_other_path():
    for ubf in ("All", "Common", "Latin"):
        assert wave_classes(rc.missions(encodings=["big5"], chars_min="4", unicode_block_filter=ubf)[0]) is None, ubf


@pytest.mark.parametrize("mi", range(len(MISSIONS)))
def test_emulated_wave_pipeline_equals_the_oracle(wave, mi):
    m = rc.missions(**MISSIONS[mi])[0]
    rng = random.Random(1000 + mi)
    for name, data in inputs(rng):
        want = oracle_findings([dict(m, mission_id=0)], data)
        for nwin, skip in ((508, 1), (60, 0), (7, 1), (123, 1)):
            got, info = emulate(wave, m, data, nwin=nwin, skip_idle=skip, swar=nwin != 60)   # (60: the class table also where ranges would do)
            assert info["bad"] == 0, (name, nwin, info)
            assert got == want, (name, nwin, skip, next(((a, b) for a, b in zip(got, want) if a != b), (len(got), len(want))))


def test_which_missions_classify_by_ranges():
    """every byte a character, the accepted ones <= 6 ranges, the accepted ones >= 0x80 of one UTF-8 length: ranges; else the table"""
    yes = [dict(encodings=["ascii"], chars_min="4"), dict(encodings=["koi8-r"], chars_min="10", unicode_block_filter="Cyrillic"),
           dict(encodings=["x-user-defined"], chars_min="4", unicode_block_filter="All"), dict(encodings=["ibm866"], chars_min="7", ascii_filter="None", unicode_block_filter="Cyrillic")]
    no = [dict(encodings=["windows-1253"], chars_min="2", unicode_block_filter="All"),   # bytes without a character
          dict(encodings=["windows-1252"], chars_min="4", unicode_block_filter="All"),   # 2- and 3-byte UTF-8 forms among the accepted
          dict(encodings=["utf-8"], chars_min="4", unicode_block_filter="0xAAAAAAA8")]   # more than six ranges of accepted lead bytes
    yes += [dict(encodings=["big5"], chars_min="10", unicode_block_filter="Cjk"), dict(encodings=["euc-kr"], chars_min="20", unicode_block_filter="Hangul")]
    no += [dict(encodings=["shift_jis"], chars_min="4", unicode_block_filter="All"),      # bytes >= 0x80 that are characters on their own
           dict(encodings=["euc-kr"], chars_min="4", unicode_block_filter="All")]        # accepted pairs of two and three UTF-8 bytes
    for kw in yes:
        assert wave_swar(rc.missions(**kw)[0]) is not None, kw
    for kw in no:
        assert wave_swar(rc.missions(**kw)[0]) is None, kw


def test_missions_the_wave_path_does_not_cover():
    for kw in (dict(encodings=["utf-8"], chars_min="4", unicode_block_filter="All", same_unicode_block=True, grep_char="32"),   # -g AND -r with more than 31 lead bytes: per buffer
               dict(encodings=["ascii"], chars_min="0"), dict(encodings=["ascii"], chars_min="70"),
               dict(encodings=["ascii"], chars_min="4", output_line_len="100"), dict(encodings=["big5"], chars_min="4", unicode_block_filter="Cjk", same_unicode_block=True),
               dict(encodings=["big5"], chars_min="4"), dict(encodings=["euc-jp"], chars_min="4", unicode_block_filter="All"), dict(encodings=["gbk"], chars_min="4")):
        assert wave_classes(rc.missions(**kw)[0]) is None, kw
