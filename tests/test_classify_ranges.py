"""The range classifiers of stage A for the alias filters (stringsext_amd/csrc/sx_classify_ranges.hpp: UTF-8 with three-byte
leads, UTF-16 with up to four unit ranges: two below U+8000, one across it, one above), compiled as host code (tests/native/classify_host.cpp) and compared byte by byte
with the decoders' rules said one position at a time: UTF-8 per encoding_rs' utf_8.rs (lead + continuation bytes, E0 / ED /
F0 / F4 narrowing the second byte), UTF-16 per utf_16.rs (units on the stream's parity, every lone surrogate an error), the
filter per reference src/mission.rs:333-348 (af bit = the ASCII code, ubf bit = the UTF-8 lead byte & 0x3F).  No GPU."""
import ctypes
import random
import zlib

import pytest

import refconfig as rc
import stringsext_amd as sx
from native.build_harness import build_classify
from test_host_logic import soup, synth

LIB = ctypes.CDLL(build_classify())
U8P = ctypes.POINTER(ctypes.c_uint8)
LIB.sxh_classify_utf8_range3.argtypes = [ctypes.c_uint32] * 6 + [ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_int, U8P, U8P]
LIB.sxh_classify_utf16_ranges.argtypes = [ctypes.POINTER(ctypes.c_uint32)] * 2 + [ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32] + [ctypes.c_int] * 3 + [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_int, U8P, U8P]


LIB.sxh_classify_product.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_uint32), ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_int, U8P, U8P]
K_UTF8_LUT, K_UTF16_LUT, K_UTF8_RANGE2, K_UTF16_RANGE, K_UTF8_RANGE3, K_UTF16_RANGES, K_UTF8_RANGE2X2 = 1, 2, 3, 4, 9, 10, 11   # csrc/sx_device.hpp ClassifierKind


def product_classifier(m, generic=False):
    """(kind, parameters) stage A runs for the Mission: the product's own choice (sx_scan_classifier)"""
    L = sx.lib()
    L.sx_scan_classifier.argtypes, L.sx_scan_classifier.restype = [ctypes.POINTER(sx.Mission), ctypes.c_int, ctypes.POINTER(ctypes.c_uint32)], ctypes.c_int
    out = (ctypes.c_uint32 * 20)()
    kind = L.sx_scan_classifier(ctypes.byref(sx.Mission.from_dict(dict(m, mission_id=m.get("mission_id", 0)))), int(generic), out)
    assert kind >= 0
    return kind, out


def lead_of(cp):
    return cp if cp < 0x80 else 0xC0 | cp >> 6 if cp < 0x800 else 0xE0 | cp >> 12 if cp < 0x10000 else 0xF0 | cp >> 18


def passes(m, lead):
    return bool(m["af"] >> lead & 1) if lead < 0x80 else bool(m["ubf"] >> (lead & 0x3F) & 1)


def naive_utf8(m, d):
    good, start = bytearray(len(d)), bytearray(len(d))
    for i, b in enumerate(d):
        if b < 0x80: n = 1
        elif 0xC2 <= b <= 0xDF: n = 2
        elif 0xE0 <= b <= 0xEF: n = 3
        elif 0xF0 <= b <= 0xF4: n = 4
        else: continue
        if i + n > len(d): continue
        if n > 1:
            lo, hi = {0xE0: (0xA0, 0xBF), 0xED: (0x80, 0x9F), 0xF0: (0x90, 0xBF), 0xF4: (0x80, 0x8F)}.get(b, (0x80, 0xBF))
            if not lo <= d[i + 1] <= hi or any(not 0x80 <= c <= 0xBF for c in d[i + 2:i + n]): continue
        if passes(m, b):
            start[i] = 1
            good[i:i + n] = b"\x01" * n
    return bytes(good), bytes(start)


def naive_utf16(m, d, be, parity):
    good, start = bytearray(len(d)), bytearray(len(d))
    unit = lambda i: d[i] << 8 | d[i + 1] if be else d[i] | d[i + 1] << 8
    for i in range(parity, len(d) - 1, 2):
        u = unit(i)
        n = 2
        if 0xD800 <= u <= 0xDBFF:
            if i + 4 > len(d) or not 0xDC00 <= unit(i + 2) <= 0xDFFF: continue
            u, n = 0x10000 + ((u & 0x3FF) << 10 | unit(i + 2) & 0x3FF), 4
        elif 0xDC00 <= u <= 0xDFFF: continue
        if passes(m, lead_of(u)):
            start[i] = 1
            good[i:i + n] = b"\x01" * n
    return bytes(good), bytes(start)


def one_range(bits, first, last):
    """(lo, hi) of the set positions if they are one run, None if empty, False otherwise — a second statement of sx_mission.cpp's test"""
    on = [i for i in range(first, last + 1) if bits >> i & 1]
    if not on: return None
    return (on[0], on[-1]) if len(on) == on[-1] - on[0] + 1 else False


def run_utf8(m, d, near):
    a, u2, u3 = one_range(m["af"], 0, 127), one_range(m["ubf"], 2, 31), one_range(m["ubf"], 32, 47)
    assert a is not False and u2 is not False and u3 and u3[0] >= 33 and not m["ubf"] >> 48 & 0x1F, "not a Mission of this classifier"
    a_lo, a_hi = a or (1, 0)
    u_lo, u_hi = (0xC0 + u2[0], 0xC0 + u2[1]) if u2 else (0x81, 0x80)
    l3_lo, l3_hi = 0xC0 + u3[0], 0xC0 + u3[1]
    ed = 0 if l3_hi < 0xED or l3_lo > 0xED else 1 if l3_hi == 0xED else 2
    out = []
    for has2, e in {(1 if u2 else 0, ed), (1, 2)}:   # the instantiation launch_scan picks, and the most general one
        good, start = (ctypes.c_uint8 * max(1, len(d)))(), (ctypes.c_uint8 * max(1, len(d)))()
        assert LIB.sxh_classify_utf8_range3(a_lo, a_hi, u_lo, u_hi, l3_lo, l3_hi, has2, e, d, len(d), near, good, start) == 0
        out.append((bytes(good[:len(d)]), bytes(start[:len(d)])))
    return out


_UNIT_RANGES = {}


def unit_ranges(m):
    """the Mission's accepted UTF-16 units as ranges (+ its high surrogates) — once per filter: 65 536 trips of Python, and run_utf16 is
    called for every input, byte order, parity and `near` (this loop was 9 of the file's 9.6 minutes)"""
    key = (m["af"], m["ubf"])
    if key not in _UNIT_RANGES:
        _UNIT_RANGES[key] = (_unit_ranges(m), [u for u in range(0xD800, 0xDC00) if passes(m, lead_of(0x10000 + ((u & 0x3FF) << 10)))])
    return _UNIT_RANGES[key]


def _unit_ranges(m):
    r = []
    for u in range(0x10000):
        if 0xD800 <= u <= 0xDFFF or not passes(m, lead_of(u)): continue
        if r and r[-1][1] + 1 == u: r[-1][1] = u
        else: r.append([u, u])
    return r


def run_utf16(m, d, be, parity, near):
    r, hs = unit_ranges(m)   # (hs: the high surrogates of the planes that pass)
    below, across, above = sum(x[1] < 0x8000 for x in r), sum(x[0] < 0x8000 <= x[1] for x in r), sum(x[0] >= 0x8000 for x in r)
    assert (r or hs) and below <= 2 and across <= 1 and above <= 1 and len(hs) == (hs[-1] - hs[0] + 1 if hs else 0), "not a Mission of this classifier"
    hs_lo, hs_hi = (hs[0], hs[-1]) if hs else (0, 0)
    lo, hi = (ctypes.c_uint32 * 6)(*[x[0] for x in r]), (ctypes.c_uint32 * 6)(*[x[1] for x in r])
    out = []
    for general in (0, 1):   # the instantiation launch_scan picks, and the one with every slot
        good, start = (ctypes.c_uint8 * max(1, len(d)))(), (ctypes.c_uint8 * max(1, len(d)))()
        assert LIB.sxh_classify_utf16_ranges(lo, hi, len(r), hs_lo, hs_hi, general, be, parity, d, len(d), near, good, start) == 0
        out.append((bytes(good[:len(d)]), bytes(start[:len(d)])))
    return out


TEXT = "中文字符串テスト한국어 텍스트ひらがなカタカナ Ελληνικά кириллица àéîõüÿĀžƀɏ̀ͯ̈Ͱ ﬁ￿퟿ꀀ ₠€ ༀ က ｶﾀｶﾅ"   # (incl. U+FFFF, U+D7FF, U+E000, U+A000)
NASTY8 = [0xE0, 0xA0, 0x9F, 0x80, 0xBF, 0xED, 0xEC, 0xEE, 0xEF, 0xE1, 0xE3, 0xE4, 0xE9, 0xEA, 0xEB, 0xF0, 0xF4, 0x90, 0x8F, 0xC2, 0xDF, 0xC1, 0x41, 0x20, 0x7F, 0x00]


def datas(rng):
    enc = lambda s: rng.choice([s.encode("utf-8", "surrogatepass"), s.encode("utf-16-le", "surrogatepass"), s.encode("utf-16-be", "surrogatepass")])
    out = [b"", b"\xe4", b"\xe4\xb8", b"\xe4\xb8\xad", b"A\xe4\xb8\xad", soup(rng, 5001), synth(rng, 20000, 1 / 100), rng.randbytes(30000),
           bytes(rng.choice(NASTY8) for _ in range(20011))]
    for _ in range(6):
        parts = []
        for _ in range(300):
            k = rng.randrange(1, 12)
            s = "".join(rng.choice(TEXT) for _ in range(k))
            parts.append(enc(s))
            parts.append(rng.choice([b"", b"\x00", b"\xff\xfe", b"\xed\xa0\x80", b"\xed\x9f\xbf", b"\xe0\x80\x80", b"\xe0\xa0\x80", b"\xe4\xb8", b"\xe9",
                                     b"\x00\xd8", b"\xd8\x00\xdc\x00", b"\x00\xd8\x00\xdc", b"\xdc\x00", b"A", b"ab c", "😀𝔘\U0010ffff\U00040000".encode("utf-16-le"),
                                     "😀\U000fffff\U00100000𝔘".encode("utf-16-be"), b"\xd8\x3d\xd8\x3d\xde\x00", b"\x3d\xd8\x3d\xd8\x00\xde", b"\xdb\xff\xdc"]) * rng.randrange(0, 3))
        out.append(b"".join(parts))
    for n in (15, 16, 17, 18, 19, 31, 32, 33, 34, 35):   # the end of the input at every phase of the 16-byte grid
        out.append(("中文字符串한국어テスト" * 4).encode("utf-8")[:n])
        out.append(("中文字符串한국어テスト" * 4).encode("utf-16-le")[:n])
        out.append(("😀𝔘😀中😀" * 4).encode("utf-16-le")[:n])
        out.append(("😀𝔘😀中😀" * 4).encode("utf-16-be")[:n])
        out.append(b"A" + ("😀𝔘😀中😀" * 4).encode("utf-16-be")[:n])
    return out


UTF8_FILTERS = ["Cjk", "Kana", "Hangul", "Asian", "0x0000ffff00000000", "0x0000fffe00000000", "0x0000c00000000000", "0x0000200000000000",
                "0x00003ffcfffffffc", "0x000003f0ffe00000", "0x0000fffefffffffc"]   # (Asian + Common; Cjk + African; everything but E0 and the 4-byte leads)


@pytest.mark.parametrize("ubf", UTF8_FILTERS)
def test_utf8_range3_equals_the_rules_byte_by_byte(ubf):
    rng = random.Random(zlib.crc32(ubf.encode()))
    for af in (None, "All", "None"):
        m = rc.missions(encodings=["utf-8"], unicode_block_filter=ubf, **({"ascii_filter": af} if af else {}))[0]
        if ubf == "0x0000ffff00000000":   # E0 among the leads: not this classifier's (sx_mission.cpp keeps the table kernel)
            with pytest.raises(AssertionError):
                run_utf8(m, b"", 0)
            continue
        for d in datas(rng):
            want = naive_utf8(m, d)
            for near in (0, 1):
                for got in run_utf8(m, d, near):
                    assert got == want, (ubf, af, len(d), near)


UTF16_FILTERS = ["Cjk", "Kana", "Hangul", "Asian", "Common", "0x0000ffff00000000", "0x00003ffcfffffffc", "0x0000800000000000", "0x0000100000000000",
                 "0x0000200000000000", "0x000000ff00000000", "0x00003800fffffffc",
                 "All", "Uncommon", "0x001f000000000000", "0x0010ffff00000000", "0x0001000000000004"]   # (... EF alone; EC; ED alone = U+D000..U+D7FF; E0..E7 = U+0800..U+7FFF; Common + Hangul; with astral planes:
                 # all / F0..F3 / every plane and no BMP character above ASCII / F4 + E0..EF / F0 + C2)


@pytest.mark.parametrize("ubf", UTF16_FILTERS)
def test_utf16_ranges_equal_the_rules_unit_by_unit(ubf):
    rng = random.Random(zlib.crc32(ubf.encode()))
    for af in (None, "All", "None"):
        m = rc.missions(encodings=["utf-16le"], unicode_block_filter=ubf, **({"ascii_filter": af} if af else {}))[0]
        for d in datas(rng):
            for be in (0, 1):
                for parity in (0, 1):
                    want = naive_utf16(m, d, be, parity)
                    for near in (0, 1):
                        for got in run_utf16(m, d, be, parity, near):
                            assert got == want, (ubf, af, len(d), be, parity, near)


# alias -> the kernel family stage A must pick (a silent fall-back to a table kernel is a 2x slowdown no parity test sees)
EXPECTED_KINDS = [
    ("utf-8", None, K_UTF8_RANGE2), ("utf-8", "African", K_UTF8_RANGE2), ("utf-8", "Common", K_UTF8_RANGE2), ("utf-8", "Cyrillic", K_UTF8_RANGE2),
    ("utf-8", "Latin", K_UTF8_RANGE2X2), ("utf-16le", "Latin", K_UTF16_RANGES), ("utf-16be", "Latin", K_UTF16_RANGES), ("utf-8", "0x00000000c00000f0", K_UTF8_RANGE2X2),
    ("utf-8", "Cjk", K_UTF8_RANGE3), ("utf-8", "Kana", K_UTF8_RANGE3), ("utf-8", "Hangul", K_UTF8_RANGE3), ("utf-8", "Asian", K_UTF8_RANGE3),
    ("utf-8", "0x00003ffcfffffffc", K_UTF8_RANGE3), ("utf-8", "All", K_UTF8_LUT), ("utf-8", "Uncommon", K_UTF8_LUT), ("utf-8", "Private", K_UTF8_LUT), ("utf-8", "0x0000800600000000", K_UTF8_LUT),   # (Misc: E1, E2, EF)
    ("utf-8", "0x0000ffff00000000", K_UTF8_LUT),
    ("utf-16le", None, K_UTF16_RANGE), ("utf-16be", "African", K_UTF16_RANGE), ("utf-16le", "Cjk", K_UTF16_RANGES), ("utf-16be", "Kana", K_UTF16_RANGES),
    ("utf-16le", "Hangul", K_UTF16_RANGES), ("utf-16be", "Asian", K_UTF16_RANGES), ("utf-16le", "0x0000800600000000", K_UTF16_RANGES), ("utf-16be", "Private", K_UTF16_RANGES), ("utf-16le", "All", K_UTF16_RANGES), ("utf-16be", "All-Asian", K_UTF16_RANGES), ("utf-8", "All-Asian", K_UTF8_LUT),
    ("utf-16be", "Uncommon", K_UTF16_RANGES), ("utf-16le", "0x0000ffff00000000", K_UTF16_RANGES), ("utf-16le", "0x0005000000000000", K_UTF16_LUT),   # (F0 and F2: two ranges of high surrogates)
]


@pytest.mark.parametrize("enc,ubf,kind", EXPECTED_KINDS)
def test_the_product_picks_the_range_kernels_and_their_parameters_classify_right(enc, ubf, kind):
    """Mission -> ClassifierKind + parameters through the C-ABI (sx_scan_classifier), then THOSE parameters through the host-compiled
    classifier, against the rules byte by byte; with generic kernels forced every Mission takes a table kernel."""
    m = rc.missions(encodings=[enc], **({"unicode_block_filter": ubf} if ubf else {}))[0]
    got_kind, params = product_classifier(m)
    assert got_kind == kind, (enc, ubf, got_kind)
    assert product_classifier(m, generic=True)[0] == (K_UTF8_LUT if enc == "utf-8" else K_UTF16_LUT)
    if kind not in (K_UTF8_RANGE3, K_UTF16_RANGES, K_UTF8_RANGE2X2):
        return
    rng = random.Random(zlib.crc32(f"{enc}{ubf}".encode()))
    be = int(enc == "utf-16be")
    for d in datas(rng):
        for parity in ((0,) if enc == "utf-8" else (0, 1)):
            want = naive_utf8(m, d) if enc == "utf-8" else naive_utf16(m, d, be, parity)
            good, start = (ctypes.c_uint8 * max(1, len(d)))(), (ctypes.c_uint8 * max(1, len(d)))()
            assert LIB.sxh_classify_product(got_kind, params, be, parity, d, len(d), 0, good, start) == 0
            assert (bytes(good[:len(d)]), bytes(start[:len(d)])) == want, (enc, ubf, len(d), parity)
