"""GPU parity tests (run with -m gpu on an MI355X).  Everything goes through the C-ABI
(libstringsext_amd.so); the oracle is only the checker."""
import os
import random

import pytest

import refconfig as rc
import stringsext_amd as sx
import sxo_binding as sxo
from golden import unit_kats as K
from product_harness import ProductScanner, run_cli_product
from test_host_logic import ENC_SETS, OPTION_SETS, soup, synth
from test_oracle_golden import CLI_CASES, rd

pytestmark = pytest.mark.gpu


def device_runs(mdict, data, parity=0, min_chars=None, generic=False, subchunk=0, capacity=0):
    sc = sx.Scanner([mdict], device=0, generic_kernels=generic, subchunk_bytes=subchunk, record_capacity=capacity)
    try:
        d = sc.alloc(len(data))
        sc.upload(d, data)
        mc = min_chars if min_chars is not None else max(1, min(mdict["chars_min_nb"], mdict["output_line_char_nb_max"]))
        got = sc.device_runs(0, d, len(data), stream_parity=parity, min_chars=mc)
        sc.free(d)
        return got, mc
    finally:
        sc.close()


def cjk_soup(rng, n):
    """CJK / Hangul / kana text in the three encodings between the byte sequences the decoders narrow (E0 80.., ED A0.., lone surrogates)."""
    text = "中文字符串テスト한국어 텍스트ひらがなカタカナ Ελληνικά àéîõüÿĀžƀȿ\u0300\u036f\u0370\u0240 ﬁ\uffff\ud7ff\ue000\ua000 ₠€ ｶﾀｶﾅ"
    junk = [b"", b"\x00", b"\xff\xfe", b"\xed\xa0\x80", b"\xed\x9f\xbf", b"\xe0\x80\x80", b"\xe0\xa0\x80", b"\xe4\xb8", b"\xe9", b"\x00\xd8",
            b"\xd8\x00\xdc\x00", b"\x00\xd8\x00\xdc", b"\xdc\x00", b"A", b"ab c", b"\xf0\x9f\x98\x80"]
    out = bytearray()
    while len(out) < n:
        s = "".join(rng.choice(text) for _ in range(rng.randrange(1, 14)))
        out += s.encode(rng.choice(["utf-8", "utf-16-le", "utf-16-be"]), "surrogatepass")
        out += rng.choice(junk) * rng.randrange(0, 3)
        if rng.random() < 0.05: out += rng.randbytes(rng.randrange(1, 200))
    return bytes(out[:n])


RUN_MISSIONS = {
    "ascii": dict(encodings=["ascii"], chars_min="4"),
    "ascii_all": dict(encodings=["ascii"], chars_min="3", unicode_block_filter="All"),
    "utf8_common": dict(encodings=["utf-8"], chars_min="10"),
    "utf8_african": dict(encodings=["utf-8"], chars_min="10", unicode_block_filter="African"),
    "utf8_all": dict(encodings=["utf-8"], chars_min="5", unicode_block_filter="All", ascii_filter="All-Ctrl+Wsp"),
    "utf8_cjk": dict(encodings=["utf-8"], chars_min="3", unicode_block_filter="Cjk"),
    "utf8_uncommon": dict(encodings=["utf-8"], chars_min="2", unicode_block_filter="Uncommon", ascii_filter="None"),
    "utf16le_african": dict(encodings=["utf-16le"], chars_min="10", unicode_block_filter="African"),
    "utf16be_common": dict(encodings=["utf-16be"], chars_min="4"),
    "utf16le_all": dict(encodings=["utf-16le"], chars_min="3", unicode_block_filter="All"),
    "utf16be_uncommon": dict(encodings=["utf-16be"], chars_min="2", unicode_block_filter="Uncommon"),
    "koi8r": dict(encodings=["koi8-r"], chars_min="4", unicode_block_filter="Cyrillic"),
    "win1251_all": dict(encodings=["windows-1251"], chars_min="6", unicode_block_filter="All"),
    "iso8859_7_greek": dict(encodings=["iso-8859-7"], chars_min="4", unicode_block_filter="Greek"),
    "win1255_hebrew": dict(encodings=["windows-1255"], chars_min="3", unicode_block_filter="Hebrew"),
    "win874_all": dict(encodings=["windows-874"], chars_min="4", unicode_block_filter="All"),
    "xmaccyr": dict(encodings=["x-mac-cyrillic"], chars_min="5", unicode_block_filter="Cyrillic"),
    # the alias filters with three-byte leads (mission.rs:167-218): Utf8Range3T / Utf16RangesT (sx_classify_ranges.hpp) in "fast" mode
    "utf8_asian": dict(encodings=["utf-8"], chars_min="3", unicode_block_filter="Asian"),
    "utf8_hangul": dict(encodings=["utf-8"], chars_min="2", unicode_block_filter="Hangul", ascii_filter="None"),
    "utf8_kana": dict(encodings=["utf-8"], chars_min="2", unicode_block_filter="Kana"),
    "utf8_common_asian": dict(encodings=["utf-8"], chars_min="4", unicode_block_filter="0x00003ffcfffffffc"),
    "utf8_e1_ef": dict(encodings=["utf-8"], chars_min="3", unicode_block_filter="0x0000fffe00000000", ascii_filter="All"),
    "utf16le_cjk": dict(encodings=["utf-16le"], chars_min="3", unicode_block_filter="Cjk"),
    "utf16be_asian": dict(encodings=["utf-16be"], chars_min="2", unicode_block_filter="Asian"),
    "utf16le_hangul": dict(encodings=["utf-16le"], chars_min="2", unicode_block_filter="Hangul", ascii_filter="None"),
    "utf16be_bmp3": dict(encodings=["utf-16be"], chars_min="3", unicode_block_filter="0x0000ffff00000000"),
    "utf16le_common_asian": dict(encodings=["utf-16le"], chars_min="4", unicode_block_filter="0x00003ffcfffffffc"),
    "utf8_latin": dict(encodings=["utf-8"], chars_min="3", unicode_block_filter="Latin"),          # two ranges of two-byte leads: Utf8Range2x2
    "utf16le_latin": dict(encodings=["utf-16le"], chars_min="3", unicode_block_filter="Latin"),    # three unit ranges below U+8000
    "utf16be_latin": dict(encodings=["utf-16be"], chars_min="2", unicode_block_filter="Latin", ascii_filter="None"),
    "odd_af": dict(encodings=["utf-8"], chars_min="4", ascii_filter="0x7ffffffe000000007ffffffe00000000"),
}


@pytest.mark.parametrize("name", sorted(RUN_MISSIONS))
@pytest.mark.parametrize("generic", [False, True], ids=["fast", "generic"])
def test_device_runs_equal_oracle_runs(name, generic):
    """Stage A: the run records of the HIP kernels == the oracle's sequential decoder."""
    m = rc.missions(**RUN_MISSIONS[name])[0]
    rng = random.Random(hash(name) & 0xFFFF)
    datas = [
        synth(rng, 200_000, 1 / 400),
        soup(rng, 70_001),
        rng.randbytes(1 << 20),
        b"A" * 5000 + rng.randbytes(3000) + "Ж".encode() * 4000 + b"\x00" * 100 + b"zz" * 3000,
        synth(rng, 1023, 1 / 50), synth(rng, 1025, 1 / 50), synth(rng, 17, 1 / 5), b"abcdefghijkl", b"",
        ("Բարեւ" * 2000).encode("utf-16-le") + b"\x41" + ("שלום" * 2000).encode("utf-16-be"),
        "𝔘𝔫𝔦𝔠𝔬𝔡𝔢😀".encode("utf-8") * 500 + "𝔘𝔫𝔦𝔠𝔬𝔡𝔢😀".encode("utf-16-le") * 500 + "𝔘𝔫𝔦😀".encode("utf-16-be") * 500,
        cjk_soup(rng, 150_001),
    ]
    for di, data in enumerate(datas):
        for parity in (0, 1):
            for sub in (1024, 4096, 65536):
                if sub != 65536 and len(data) > 300_000:
                    continue
                got, mc = device_runs(m, data, parity=parity, generic=generic, subchunk=sub)
                want = sxo.runs(m, data, stream_parity=parity, min_chars=mc)
                assert got == want, (name, di, parity, sub, len(got), len(want))


@pytest.mark.parametrize("n", [13, 14, 15, 16, 17, 20])
def test_long_thresholds_with_characters_across_lane_edges(n):
    """The candidate test of the scan kernels works on the lane's 16 bytes above the 16 before them, and takes the lower half
    without what ITS predecessor's last character spilled into it (up to three continuation bytes): stretches of exactly
    n .. n+2 characters of 2, 3 and 4 bytes at every phase of the 16-byte grid, for thresholds around the window's reach."""
    rng = random.Random(100 + n)
    words = ["Ж", "中", "字", "😀", "𝔘", "é", "א"]
    parts = []
    for _ in range(6000):
        k = rng.choice([n - 1, n, n, n + 1, n + 2])
        txt = "".join(rng.choice(words[:rng.choice([1, 3, 5, 7])]) for _ in range(k))
        parts.append(rng.choice([b"\xff", b"\x00\x01", b"\x80", b"\xc0\xff\xfe", b"\n\n\n\n\n"]) * rng.randrange(1, 4))
        parts.append(rng.choice([txt.encode("utf-8"), txt.encode("utf-16-le"), txt.encode("utf-16-be")]))
    data = b"".join(parts)
    for flags in (dict(encodings=["utf-8"], unicode_block_filter="All"), dict(encodings=["utf-16le"], unicode_block_filter="All"),
                  dict(encodings=["utf-16be"], unicode_block_filter="All"), dict(encodings=["utf-8"], unicode_block_filter="Cyrillic")):
        m = rc.missions(chars_min=str(n), **flags)[0]
        for parity in (0, 1):
            for generic in (False, True):
                got, mc = device_runs(m, data, parity=parity, generic=generic, subchunk=4096)
                assert got == sxo.runs(m, data, stream_parity=parity, min_chars=mc), (n, flags, parity, generic)
        ms = rc.missions(chars_min=str(n), **flags)
        assert run_cli_product(ms, [data], radix="x", device=0) == sxo.run_cli(ms, [data], radix="x"), (n, flags)


def fused_runs(mdicts, data, parity=0, subchunk=0, min_chars=None):
    """Stage A of several Missions in one call (sx_device_runs_multi): -> run lists, thresholds, fused_mask."""
    sc = sx.Scanner(mdicts, device=0, subchunk_bytes=subchunk)
    try:
        d = sc.alloc(len(data))
        sc.upload(d, data)
        mc = min_chars or [max(1, min(m["chars_min_nb"], m["output_line_char_nb_max"])) for m in mdicts]
        got = sc.device_runs_multi(list(range(len(mdicts))), d, len(data), stream_parity=parity, min_chars=mc)
        mask = sc.stats().fused_mask
        sc.free(d)
        return got, mc, mask
    finally:
        sc.close()


FUSED_SETS = {
    # BASELINE's headline Missions: all three in one launch, the UTF-16 ones behind the prefilter (n >= 7)
    "c3": (dict(encodings=["utf-8", "utf-16le", "utf-16be"], chars_min="10", unicode_block_filter="African"), 0b111),
    # thresholds 3 .. 6: the prefilter on aligned PAIRS of high bytes (the reference's default -n 4); below 3: the UTF-16 slots classify every tile
    "n4": (dict(encodings=["utf-8", "utf-16le", "utf-16be"], chars_min="4", unicode_block_filter="Common"), 0b111),
    "n3": (dict(encodings=["utf-16le", "utf-16be", "utf-8"], chars_min="3", unicode_block_filter="Hebrew"), 0b111),
    "n2": (dict(encodings=["utf-8", "utf-16be", "utf-16le"], chars_min="2", unicode_block_filter="Armenian", ascii_filter="None"), 0b111),
    "n7_mixed": (dict(encodings=["utf-16be,7,,Armenian", "utf-8,3,,Greek", "utf-16le,12,,Hebrew"]), 0b111),   # slots in another order, a filter per Mission
    "pairs": (dict(encodings=["utf-16le", "utf-8"], chars_min="8", unicode_block_filter="Arabic"), 0b11),
    "le_be": (dict(encodings=["utf-16le", "utf-16be"], chars_min="9", unicode_block_filter="Cyrillic"), 0b11),
    "ascii_only_af": (dict(encodings=["utf-8", "utf-16be"], chars_min="7", unicode_block_filter="None"), 0b11),
    # a Mission the fused kernel has no slot for keeps its own launch next to the fused pair
    "with_koi8r": (dict(encodings=["utf-8,,,African", "koi8-r,,,Cyrillic", "utf-16le,,,African"], chars_min="10"), 0b101),
    # two Missions for the same slot: the second one keeps its own launch
    "two_utf8": (dict(encodings=["utf-8,,,African", "utf-8,,,Greek", "utf-16be,,,Greek"], chars_min="10"), 0b101),
}


def utf16_dense(rng, n):
    """What the UTF-16 prefilter has to get right: stretches of accepted units of every length around 7 .. 40 at every byte phase, across
    tile (1 KiB) and sub-chunk edges, between bytes whose high-byte positions pass or fail the prefilter's mask."""
    words = ["Բարեւ", "שלום", "مرحبا", "abc XYZ", "Жук", "Ελλάς", "/usr/lib"]
    out = bytearray()
    while len(out) < n:
        k = rng.choice([1, 3, 6, 7, 8, 10, 13, 14, 20, 40, 600])
        txt = "".join(rng.choice(words) for _ in range(k))[:k]
        out += txt.encode(rng.choice(["utf-16-le", "utf-16-be", "utf-8"]))
        r = rng.random()
        if r < 0.3: out += rng.randbytes(rng.randrange(0, 9))
        elif r < 0.5: out += b"\x00" * rng.randrange(0, 2100)          # high bytes that pass: the prefilter must not matter
        elif r < 0.7: out += b"\xff" * rng.randrange(0, 2100)          # ... that fail: skipped tiles between stretches
        elif r < 0.8: out += bytes([rng.randrange(8), rng.randrange(256)]) * rng.randrange(0, 600)
        else: out += rng.randbytes(rng.randrange(900, 1200))
    return bytes(out[:n])


@pytest.mark.parametrize("name", sorted(FUSED_SETS))
def test_fused_scan_runs_equal_oracle_runs(name):
    """The fused stage A (one read of the buffer for several Missions, sx_fused.hip) reports, per Mission, the oracle's runs — and did fuse."""
    flags, want_mask = FUSED_SETS[name]
    ms = rc.missions(**flags)
    rng = random.Random(hash(name) & 0xFFFF)
    datas = [
        utf16_dense(rng, 300_001), utf16_dense(rng, 70_000),
        synth(rng, 200_000, 1 / 400), soup(rng, 70_001), rng.randbytes(1 << 20),
        b"A" * 5000 + rng.randbytes(3000) + "Ж".encode() * 4000 + b"\x00" * 100 + b"zz" * 3000,
        ("Բարեւ" * 2000).encode("utf-16-le") + b"\x41" + ("שלום" * 2000).encode("utf-16-be"),
        b"\x00" * 3000 + ("a\x00" * 7).encode("latin-1") + b"\xff" * 5000 + ("\x00b" * 7).encode("latin-1") + b"\xff" * 1017 + ("c\x00" * 9).encode("latin-1"),
        synth(rng, 1023, 1 / 50), synth(rng, 1025, 1 / 50), synth(rng, 2049, 1 / 50), synth(rng, 17, 1 / 5), b"abcdefghijkl", b"",
        cjk_soup(rng, 100_001),
    ]
    for di, data in enumerate(datas):
        for parity in (0, 1):
            for sub in (1024, 4096, 65536):
                if sub != 65536 and len(data) > 350_000:
                    continue
                got, mc, mask = fused_runs(ms, data, parity=parity, subchunk=sub)
                if data:
                    assert mask == want_mask, (name, di, bin(mask))
                for k, m in enumerate(ms):
                    want = sxo.runs(m, data, stream_parity=parity, min_chars=mc[k])
                    assert got[k] == want, (name, di, parity, sub, k, len(got[k]), len(want))


def test_device_runs_min_chars_sweep_and_overflow():
    m = rc.missions(encodings=["ascii"], chars_min="4")[0]
    data = random.Random(3).randbytes(1 << 20)
    for mc in (1, 2, 3, 9, 16, 17, 18, 40, 255):
        got, _ = device_runs(m, data, min_chars=mc, capacity=1024)  # forces the grow-and-rerun path for small mc
        assert got == sxo.runs(m, data, min_chars=mc), mc


@pytest.mark.parametrize("expected,flags,inputs", CLI_CASES, ids=[c[0] for c in CLI_CASES])
@pytest.mark.parametrize("chunk", [None, 8192])
@pytest.mark.parametrize("generic", [False, True], ids=["fast", "generic"])
def test_cli_golden_outputs_on_gpu(expected, flags, inputs, chunk, generic):
    out = run_cli_product(rc.missions(**flags), [rd(i) for i in inputs], radix="x", chunk_bytes=chunk, device=0,
                          generic_kernels=generic)
    assert out == rd(expected)


@pytest.mark.parametrize("kat", K.SCAN_KATS, ids=[k["name"] for k in K.SCAN_KATS])
def test_scan_known_answers_on_gpu(kat):
    sc = ProductScanner(kat["mission"], device=0)
    for i, call in enumerate(kat["calls"]):
        got = sc.scan(call["input"], file_id=0, is_last=call["is_last"])
        slim = [dict(position=f["position"], precision=f["precision"], s=f["s"]) for f in got]
        if "findings" in call:
            assert slim == call["findings"], (kat["name"], i)
        if "findings_prefix" in call:
            assert slim[:len(call["findings_prefix"])] == call["findings_prefix"], (kat["name"], i)


def _cases():
    rng = random.Random(77)
    out = []
    for i, opts in enumerate(OPTION_SETS):
        for j, encs in enumerate(ENC_SETS):
            if (i + 2 * j) % 4 == 0:
                out.append((opts, encs, rng.randrange(1 << 30)))
    return out


@pytest.mark.parametrize("opts,encs,seed", _cases(), ids=lambda v: str(v)[:40])
def test_end_to_end_equals_oracle(opts, encs, seed):
    """HIP kernels + host replay == the oracle's full scan, byte for byte."""
    rng = random.Random(seed)
    ms = rc.missions(encodings=encs, **opts)
    files = [synth(rng, 150_000, 1 / 300), soup(rng, 20_001), synth(rng, 4096 * 3 + 1, 1 / 100), b"",
             synth(rng, 50_000, 1 / 2000)]
    want = sxo.run_cli(ms, files, radix="x")
    for chunk, sub in ((None, 0), (16384, 1024)):
        got = run_cli_product(ms, files, radix="x", chunk_bytes=chunk, device=0, subchunk_bytes=sub)
        assert got == want, (chunk, sub)


def test_device_resident_input_and_background_generator():
    """sx_scan_device on HBM-resident synthetic background (BASELINE.md §3 generator on the
    device) == the oracle on the same bytes generated on the host."""
    ms = rc.missions(encodings=["utf-8", "utf-16le", "utf-16be"], chars_min="10", unicode_block_filter="African")
    n = (8 << 20) + 4096 * 3 + 17
    sc = sx.Scanner(ms, device=0)
    d = sc.alloc(n)
    sc.fill_background(d, 12345, n)
    host = sxo.background(12345, n)
    assert sc.download(d, n) == host
    res = sc.scan_device(d, n, file_id=1)
    got = sx.OUTPUT_BOM + res.printed(n_inputs=1, radix="x") + b"\n"
    st = sc.stats()
    assert got == sxo.run_cli(ms, [host], radix="x")
    assert st.bytes_scanned == 3 * n and st.replay_bytes < 0.1 * 3 * n
    # C2: default UBF, one mission
    ms2 = rc.missions(encodings=["utf-8"], chars_min="10")
    sc2 = sx.Scanner(ms2, device=0)
    res2 = sc2.scan_device(d, n, file_id=1)  # pointer from another context on the same device
    assert sx.OUTPUT_BOM + res2.printed(n_inputs=1, radix="x") + b"\n" == sxo.run_cli(ms2, [host], radix="x")
    sc.free(d)
    sc.close(); sc2.close()


def test_large_input_properties():
    """At a size the oracle cannot cover in seconds: invariances the path must have."""
    ms = rc.missions(encodings=["utf-8", "utf-16le", "utf-16be"], chars_min="10", unicode_block_filter="African")
    n = 1 << 30
    sc = sx.Scanner(ms, device=0)
    d = sc.alloc(n)
    sc.fill_background(d, 0, n)
    r1 = sc.scan_device(d, n, file_id=1)
    a = r1.findings()
    sc.reset()
    # pieces (scan kernels queued two deep) must not matter
    import os
    os.environ["SX_PIECE_MIB"] = "96"
    try:
        rp = sc.scan_device(d, n, file_id=1)
        assert len(rp.segments()) > 1 and rp.findings() == a
        rp.free()
    finally:
        del os.environ["SX_PIECE_MIB"]
    sc.reset()
    # sub-chunk size must not matter; neither must scanning the same bytes again
    sc_b = sx.Scanner(ms, device=0, subchunk_bytes=1 << 20)
    b = sc_b.scan_device(d, n, file_id=1).findings()
    assert a == b
    # two chunks with carried state == one chunk
    half = n // 2
    import ctypes
    c1 = sc.scan_device(d, half, file_id=1).findings()
    c2 = sc.scan_device(ctypes.c_void_p(d.value + half), n - half, file_id=1).findings()
    for f in c2:
        f["slice_index"] += half // 4096
    assert c1 + c2 == a
    # oracle on a 64 MiB prefix
    pre = 64 << 20
    host = sxo.background(0, pre)
    want = sxo.run_cli(ms, [host], radix="x")
    sc.reset()
    got = sx.OUTPUT_BOM + sc.scan_device(d, pre, file_id=1).printed(radix="x") + b"\n"
    assert got == want
    # positions sorted within a mission, all inside the input
    for mid in range(3):
        pos = [f["position"] for f in a if f["mission_id"] == mid]
        assert pos == sorted(pos) and (not pos or pos[-1] < n)
    sc.free(d)
    sc.close(); sc_b.close()


@pytest.mark.parametrize("name", ["ascii", "utf8_common", "utf16le_african", "koi8r"])
def test_device_join_equals_oracle_runs(name, monkeypatch):
    """Records sorted and joined into runs on the device (rocPRIM sort/scan/select, the path large
    buffers take) == the oracle's runs; forced here for small buffers, with tiny sub-chunks so
    that runs are chained across many records."""
    monkeypatch.setenv("SX_DEVICE_JOIN_MIN", "1")
    m = rc.missions(**RUN_MISSIONS[name])[0]
    rng = random.Random(77)
    datas = [rng.randbytes(3 << 20), b"A" * 300_000 + rng.randbytes(5000) + b"B" * 70_000,
             synth(rng, 500_000, 1 / 300), ("ab" * 100_000).encode("utf-16-le"), b"x"]
    for data in datas:
        for sub in (1024, 65536):
            got, mc = device_runs(m, data, parity=0, subchunk=sub)
            want = sxo.runs(m, data, stream_parity=0, min_chars=mc)
            assert got == want, (name, len(data), sub, len(got), len(want))


def dense(rng, n, max_gap, alphabet):
    """Strings of 3..70 chars packed with short gaps: replay regions run into each other all the time."""
    out = bytearray()
    while len(out) < n:
        k = rng.randrange(3, 70 if rng.random() < 0.9 else 400)
        out += "".join(rng.choice(alphabet) for _ in range(k)).encode("utf-8")
        out += bytes(rng.choice(b"\x00\x01\x7f\xff\xc0\x80") for _ in range(rng.randrange(1, max_gap)))
    return bytes(out[:n])


@pytest.mark.parametrize("max_gap", [3, 40, 200, 1500])
@pytest.mark.parametrize("opts", [dict(chars_min="4"), dict(chars_min="12", output_line_len="32"),
                                  dict(chars_min="5", unicode_block_filter="Cyrillic", ascii_filter="None")],
                         ids=["n4", "n12q32", "cyr"])
def test_dense_strings_device_replay_and_stitch(max_gap, opts, monkeypatch):
    """Stage B on the device (regions, which of them stand, output offsets, entry part splice)
    == the oracle, on inputs where regions chain and overrun each other constantly."""
    rng = random.Random(max_gap * 7 + len(opts))
    alphabet = "abcdefghij XYZ019_-éжЖдяבשλ€"
    data = dense(rng, 3_000_000 + rng.randrange(5000), max_gap, alphabet)
    ms = rc.missions(encodings=["utf-8"], **opts)
    want = sxo.run_cli(ms, [data], radix="x")
    for chunk in (None, 1 << 20):
        got = run_cli_product(ms, [data], radix="x", chunk_bytes=chunk, device=0, device_replay=True)
        assert got == want, (max_gap, chunk)
    for slabs, cap in (("3", None), ("7", None), ("16", "1"), ("4", "2")):   # the mission replayed in slabs (also with regions given back)
        monkeypatch.setenv("SX_SLABS", slabs)
        if cap: monkeypatch.setenv("SX_MAX_REGION_WINDOWS", cap)
        for chunk in (None, 1 << 20):
            assert run_cli_product(ms, [data], radix="x", chunk_bytes=chunk, device=0, device_replay=True) == want, (max_gap, slabs, chunk)
    # small chunks: a slab's last region often runs to the chunk's end, the state handed on is the replay of THAT region
    monkeypatch.setenv("SX_DEVICE_JOIN_MIN", "1")
    small = data[:400_000]
    assert run_cli_product(ms, [small], radix="x", chunk_bytes=4096, device=0, device_replay=True) == sxo.run_cli(ms, [small], radix="x")
    monkeypatch.delenv("SX_DEVICE_JOIN_MIN")
    monkeypatch.delenv("SX_SLABS"); monkeypatch.delenv("SX_MAX_REGION_WINDOWS")
    monkeypatch.setenv("SX_HOST_STITCH", "1")
    assert run_cli_product(ms, [data], radix="x", device=0, device_replay=True) == want


@pytest.mark.parametrize("q", ["100", "255", "65"])
def test_long_output_lines_replay_on_the_device(q, monkeypatch):
    """64 < q <= 255 (windows of up to 510 bytes): the replay kernels' QBIG instantiations (round 4; the host replayed these before) —
    dense and sparse input, three decoder families, chunks, slabs"""
    rng = random.Random(int(q))
    alphabet = "abcdefghij XYZ019_-éжЖдяבשλ€"
    data = dense(rng, 1_500_000, 40, alphabet) + dense(rng, 500_000, 3, alphabet) + synth(rng, 1_000_000, 1 / 300)
    for flags in (dict(encodings=["utf-8"], chars_min="4", unicode_block_filter="All"), dict(encodings=["utf-16le", "koi8-r"], chars_min="3"),
                  dict(encodings=["big5", "ascii"], chars_min="70", unicode_block_filter="All")):
        ms = rc.missions(output_line_len=q, **flags)
        want = sxo.run_cli(ms, [data], radix="x")
        for chunk, slabs in ((None, None), (1 << 20, "3"), (65536, None)):
            if slabs: monkeypatch.setenv("SX_SLABS", slabs)
            else: monkeypatch.delenv("SX_SLABS", raising=False)
            assert run_cli_product(ms, [data], radix="x", chunk_bytes=chunk, device=0, device_replay=True) == want, (q, flags, chunk)
    monkeypatch.delenv("SX_SLABS", raising=False)


@pytest.mark.parametrize("device_replay", [None, True, False])
def test_pieces_pipeline_equals_oracle(device_replay, monkeypatch):
    """A large buffer is scanned piece by piece with the scan kernels queued two deep
    (scan_common); forced here with 1 MiB pieces.  Host-resident and device-resident input,
    several missions with findings (k-merge per piece), strings crossing piece edges."""
    monkeypatch.setenv("SX_PIECE_MIB", "1")
    rng = random.Random(4242)
    ms = rc.missions(encodings=["utf-8", "utf-16le", "ascii"], chars_min="5")
    data = bytearray(synth(rng, (9 << 20) + 4096 * 5 + 123, 1 / 700))
    for edge in range(1 << 20, len(data), 1 << 20):   # strings over the piece edges
        data[edge - 40:edge + 40] = ("edge%08d-" % edge).encode() * 6 + b"01234567"
        data[edge + 4096 - 3:edge + 4096 + 9] = "żółw-żółw".encode("utf-16-le")[:12]
    data = bytes(data)
    want = sxo.run_cli(ms, [data], radix="x")
    got = run_cli_product(ms, [data], radix="x", device=0, device_replay=device_replay)
    assert got == want
    sc = sx.Scanner(ms, device=0, device_replay=device_replay)
    d = sc.alloc(len(data)); sc.upload(d, data)
    res = sc.scan_device(d, len(data), file_id=1)
    assert len(res.segments()) > 1
    assert sx.OUTPUT_BOM + res.printed(n_inputs=1, radix="x") + b"\n" == want
    seg_view = res.findings()
    fb, arena = res.raw()           # joins the segments
    assert len(res.segments()) == 1 and res.findings() == seg_view and len(fb) == len(seg_view) * 32
    sc.free(d); sc.close()


def test_ingest_pipeline_file_and_stream(tmp_path, monkeypatch):
    """sx_scan_file / sx_scan_stream: reader thread + pinned double buffer + H2D overlapped with the
    scan; every chunk is one sx_scan call with carried state, so the printed text equals the oracle's."""
    rng = random.Random(31)
    ms = rc.missions(encodings=["utf-8", "utf-16le", "ascii"], chars_min="6")
    data = synth(rng, (5 << 20) + 4096 * 3 + 77, 1 / 500)
    want = sxo.run_cli(ms, [data], radix="x")
    path = tmp_path / "image.bin"
    path.write_bytes(data)
    for chunk, mapped in ((1 << 20, False), (64 << 10, False), (0, False), (1 << 20, True)):
        if mapped:
            monkeypatch.setenv("SX_INGEST_MMAP", "1")
        sc = sx.Scanner(ms, device=0)
        parts = sc.scan_file(str(path), chunk_bytes=chunk, file_id=1)
        got = sx.OUTPUT_BOM + b"".join(r.printed(n_inputs=1, radix="x") for r in parts) + b"\n"
        assert got == want, chunk
        assert len(parts) == (1 if chunk == 0 else -(-len(data) // chunk))
        sc.close()
    # a Python reader that returns short reads
    sc = sx.Scanner(ms, device=0)
    pos = [0]

    def readinto(view):
        n = min(len(view), rng.randrange(1, 300_000), len(data) - pos[0])
        view[:n] = data[pos[0]:pos[0] + n]
        pos[0] += n
        return n
    parts = sc.scan_stream(readinto, chunk_bytes=2 << 20, file_id=1)
    assert sx.OUTPUT_BOM + b"".join(r.printed(n_inputs=1, radix="x") for r in parts) + b"\n" == want
    sc.close()


SWITCHES = [
    {}, {"SX_REGION_CAP": "2"}, {"SX_REGION_CAP": "0"}, {"SX_HOST_STITCH": "1"}, {"SX_NO_REPLAY_SKIP": "1"},
    {"SX_NO_REPLAY_CACHE": "1"}, {"SX_REPLAY_CACHE_MIB": "0"}, {"SX_HOST_MERGE": "1"}, {"SX_MISSION_STREAMS": "1"}, {"SX_DEVICE_JOIN_MIN": "1"},
    {"SX_SCAN_BLOCKS_PER_CU": "3", "SX_SCAN_CUS": "2"}, {"SX_PIECE_MIB": "2", "SX_REGION_CAP": "4"},
    {"SX_REGION_CAP": "1"}, {"SX_REGION_CAP": "2", "SX_NO_LARGE_REGIONS": "1"}, {"SX_STITCH_BLOCK": "512"}, {"SX_STITCH_BLOCK": "5"},
    {"SX_MAX_REGION_WINDOWS": "2"}, {"SX_DEFER_MIN_BYTES": "1"},
]


@pytest.mark.parametrize("env", SWITCHES, ids=lambda e: "+".join(f"{k[3:]}={v}" for k, v in e.items()) or "default")
def test_every_switch_gives_the_same_text(env, monkeypatch):
    """Each alternative path behind an environment switch — record pool vs regions and
    the overflow fallback between them, host vs device for the join, the stitch and the merge, replay
    with and without shortcuts / output cache, stream layout, persistent grid, pieces — must print
    what the oracle prints, on a planted image where all three missions have findings."""
    from test_gpu_baseline_configs import planted_image
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    ms = rc.missions(encodings=["utf-8", "utf-16le", "utf-16be"], chars_min="10", unicode_block_filter="African")
    img = planted_image(12 << 20, 11, every=8192)[:12 << 20]
    want = sxo.run_cli(ms, [img], radix="x")
    sc = sx.Scanner(ms, device=0, device_replay=True)
    d = sc.alloc(len(img)); sc.upload(d, img)
    for _ in range(2):      # twice: the second call starts from what the first learnt (mission order, dense flags)
        sc.reset()
        res = sc.scan_device(d, len(img), file_id=1)
        assert sx.OUTPUT_BOM + res.printed(n_inputs=1, radix="x") + b"\n" == want
        res.free()
    sc.free(d); sc.close()


@pytest.mark.parametrize("env", [{}, {"SX_DEFER_MIN_BYTES": "1"}, {"SX_MERGE_PART_FINDINGS": "20000"},
                                 {"SX_DEFER_MIN_BYTES": "1", "SX_MERGE_PART_FINDINGS": "7000", "SX_MERGE_PART_MIB": "1"},
                                 {"SX_DEFER_MIN_BYTES": "100000", "SX_MERGE_PART_FINDINGS": "1024"},
                                 {"SX_SEQ_PIECE_KIB": "512"}, {"SX_SEQ_PIECE_KIB": "640", "SX_DEFER_MIN_BYTES": "1"},
                                 {"SX_SEQ_PIECE_KIB": "300", "SX_DEFER_MIN_BYTES": "1", "SX_MERGE_PART_FINDINGS": "7000"},
                                 {"SX_SEQ_PIECE_KIB": "1024", "SX_DEFER_MIN_BYTES": "1", "SX_WAVE_REPLAY": "1"}],
                         ids=lambda e: "+".join(f"{k[3:]}={v}" for k, v in e.items()) or "default")
def test_device_merge_in_parts(env, monkeypatch):
    """The merger on the device (sx_stage_b.cpp device_merge): several missions with many findings each, their output held
    back on the device (SX_DEFER_MIN_BYTES) or already copied, interleaved in one part or in many (one result segment each);
    SX_SEQ_PIECE_KIB: the buffer in pieces scanned one after the other, a piece's findings copied while the next piece is
    scanned (scan_common's sequential pieces, as for a buffer whose output is gigabytes)."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    ms = rc.missions(encodings=["ascii", "utf-8", "windows-1252", "utf-16le"], chars_min="4")
    data = sxo.background(77, 3 << 20) + bytes(random.Random(5).choices(b"abcdefgh \x00\xc3\xa9", k=1 << 20))
    want = sxo.run_cli(ms, [data], radix="x")
    sc = sx.Scanner(ms, device=0, device_replay=True)
    d = sc.alloc(len(data)); sc.upload(d, data)
    for _ in range(2):
        sc.reset()
        res = sc.scan_device(d, len(data), file_id=1)
        if env.get("SX_MERGE_PART_FINDINGS") or env.get("SX_SEQ_PIECE_KIB"):
            assert len(res.segment_pointers()) > 1
        assert (sc.stats().seq_pieces >= 4) == bool(env.get("SX_SEQ_PIECE_KIB"))
        assert sx.OUTPUT_BOM + res.printed(n_inputs=1, radix="x") + b"\n" == want
        res.free()
    sc.free(d); sc.close()


@pytest.mark.parametrize("flush", [False, True])
def test_large_host_buffers_take_the_ingest_pipeline(flush, monkeypatch):
    """sx_scan of a large host buffer = pinned staging + chunks copied while the chunk before is
    scanned, ONE result with a segment per chunk (slice indices running on); forced here from 2 MiB
    on.  Buffer sizes around the chunk grid, and the is_last flush on the final chunk."""
    monkeypatch.setenv("SX_SCAN_STREAM_MIB", "1")
    rng = random.Random(8)
    ms = rc.missions(encodings=["utf-8", "utf-16be", "ascii"], chars_min="5")
    for n in ((5 << 20) + 4096 * 2 + 33, 4 << 20, (4 << 20) + 1, (3 << 20) - 1):
        data = synth(rng, n, 1 / 300) [:n - 20] + b"tail string without end"[:20]
        want = sxo.run_cli(ms, [data], radix="x", flush_at_eof=flush)
        got = run_cli_product(ms, [data], radix="x", device=0, flush_at_eof=flush)
        assert got == want, (n, flush)
    sc = sx.Scanner(ms, device=0)
    res = sc.scan(data, file_id=1)
    assert len(res.segments()) >= 2
    idx = [f["slice_index"] for f in res.findings()]
    assert idx == sorted(idx) and idx[-1] == (len(data) - 1) // 4096 or idx[-1] <= (len(data) - 1) // 4096
    sc.close()
