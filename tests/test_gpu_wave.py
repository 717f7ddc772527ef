"""GPU parity of the wave-cooperative stage B (csrc/sx_wave_dev.hip, sx_wave.cpp) — run with -m gpu on an MI355X.
Everything goes through the C-ABI; the oracle is only the checker.  SX_WAVE_REPLAY=1 makes every buffer of a covered
Mission take the wave kernels (by default only string-dense buffers do)."""
import os
import random

import pytest

import refconfig as rc
import stringsext_amd as sx
import sxo_binding as sxo
from product_harness import run_cli_product
from test_host_logic import soup, synth
from test_wave_core import MISSIONS, inputs, records, text_lines

pytestmark = pytest.mark.gpu


@pytest.fixture
def wave_forced():
    old = {k: os.environ.get(k) for k in ("SX_WAVE_REPLAY", "SX_WAVE_BATCHES")}
    os.environ["SX_WAVE_REPLAY"] = "1"
    yield
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


def wave_windows_of_a_scan(ms, data):
    sc = sx.Scanner(ms, device=0)
    try:
        res = sc.scan(data, file_id=1)
        n = sc.stats().wave_windows
        res.free()
        return n
    finally:
        sc.close()


@pytest.mark.parametrize("mi", range(len(MISSIONS)))
def test_wave_path_equals_the_oracle(wave_forced, mi):
    ms = rc.missions(**MISSIONS[mi])
    rng = random.Random(2000 + mi)
    for name, data in inputs(rng):
        want = sxo.run_cli(ms, [data], radix="x")
        for chunk, batches in ((None, None), (16384, "1"), (8192, "3")):
            if batches:
                os.environ["SX_WAVE_BATCHES"] = batches
            else:
                os.environ.pop("SX_WAVE_BATCHES", None)
            if batches == "3":
                os.environ["SX_WAVE_LUT"] = "1"   # the class table also where the kernels would classify by ranges
            got = run_cli_product(ms, [data], radix="x", device=0, chunk_bytes=chunk)
            os.environ.pop("SX_WAVE_LUT", None)
            assert got == want, (name, chunk, batches)
        if len(data) >= 8192:
            assert wave_windows_of_a_scan(ms, data) > 0, name   # the wave kernels did run


def test_both_writers_give_the_same(wave_forced):
    """the lane-per-finding writer (descriptors left by the count pass, the default), the window-parallel writer (SX_WAVE_DESC=0), and the
    way from one to the other when a wavefront finds more than its descriptors hold (SX_WAVE_DESC_CAP) — in slabs too"""
    rng = random.Random(77)
    cases = [(dict(encodings=["ascii"], chars_min="4"), rng.randbytes(700_000) + text_lines(rng, 300_000)),
             (dict(encodings=["utf-8"], chars_min="3"), soup(rng, 400_000) + text_lines(rng, 200_000, 100, 700)),
             (dict(encodings=["koi8-r"], chars_min="2", unicode_block_filter="Cyrillic"), rng.randbytes(600_000)),
             (dict(encodings=["big5"], chars_min="3", unicode_block_filter="Cjk"), rng.randbytes(500_000) + text_lines(rng, 100_000)),
             (dict(encodings=["shift_jis"], chars_min="10"), rng.randbytes(300_000) + "日本語のテキスト、\n".encode("shift_jis") * 9000)]
    keys = ("SX_WAVE_DESC", "SX_WAVE_DESC_CAP", "SX_WAVE_SLABS")
    try:
        for flags, data in cases:
            ms = rc.missions(**flags)
            want = sxo.run_cli(ms, [data], radix="x")
            for sw in ({}, {"SX_WAVE_DESC": "0"}, {"SX_WAVE_DESC_CAP": "5"}, {"SX_WAVE_DESC_CAP": "300", "SX_WAVE_SLABS": "3"}, {"SX_WAVE_SLABS": "4"}):
                for k in keys:
                    os.environ.pop(k, None)
                os.environ.update(sw)
                for chunk in (None, 1 << 17):
                    assert run_cli_product(ms, [data], radix="x", device=0, chunk_bytes=chunk) == want, (flags, sw, chunk)
        # the overflow is seen and counted
        os.environ["SX_WAVE_DESC_CAP"] = "5"; os.environ.pop("SX_WAVE_SLABS", None); os.environ.pop("SX_WAVE_DESC", None)
        sc = sx.Scanner(rc.missions(encodings=["ascii"], chars_min="4"), device=0)
        res = sc.scan(rng.randbytes(1 << 20), file_id=1)
        assert sc.stats().wave_desc_overflows > 0
        res.free(); sc.close()
    finally:
        for k in keys:
            os.environ.pop(k, None)


def test_wave_path_next_to_other_missions(wave_forced):
    """three Missions, two of them through the wave kernels: merge order, per-Mission state from chunk to chunk, two files"""
    ms = rc.missions(encodings=["ascii", "utf-8", "koi8-r,,,Cyrillic"], chars_min="5")
    rng = random.Random(7)
    files = [text_lines(rng, 150_000) + synth(rng, 100_000, 1 / 80), rng.randbytes(70_001), b"", text_lines(rng, 30_000, 100, 600)]
    want = sxo.run_cli(ms, files, radix="x")
    for chunk in (None, 65536, 8192):
        assert run_cli_product(ms, files, radix="x", device=0, chunk_bytes=chunk) == want, chunk
    os.environ["SX_WAVE_THREADS"] = "0"      # the wave Missions one after the other on the shared stream (default: a host thread and a stream each)
    try:
        for chunk in (65536, 8192):
            assert run_cli_product(ms, files, radix="x", device=0, chunk_bytes=chunk) == want, chunk
    finally:
        del os.environ["SX_WAVE_THREADS"]
    os.environ["SX_DEFER_MIN_BYTES"] = "1"   # the Missions' outputs stay in HBM and are interleaved there
    try:
        assert run_cli_product(ms, files, radix="x", device=0) == want
    finally:
        os.environ.pop("SX_DEFER_MIN_BYTES", None)


def test_dense_buffers_take_the_wave_path_by_default():
    rng = random.Random(3)
    ms = rc.missions(encodings=["ascii"], chars_min="4")
    data = rng.randbytes(1 << 22)
    assert run_cli_product(ms, [data], radix="x", device=0) == sxo.run_cli(ms, [data], radix="x")
    assert wave_windows_of_a_scan(ms, data) > 0
    sparse = rc.missions(encodings=["ascii"], chars_min="40")
    assert wave_windows_of_a_scan(sparse, data) == 0   # few runs: the lane-per-region path
    os.environ["SX_WAVE_REPLAY"] = "0"
    try:
        assert wave_windows_of_a_scan(ms, data) == 0
        assert run_cli_product(ms, [data], radix="x", device=0) == sxo.run_cli(ms, [data], radix="x")
    finally:
        os.environ.pop("SX_WAVE_REPLAY", None)


def test_wave_path_large_and_text(wave_forced):
    """64 MiB of text and of the synthetic background: full text diff against the oracle"""
    rng = random.Random(11)
    blob = text_lines(rng, 1 << 20)
    text = (blob * 64)[:64 << 20]
    for ms in (rc.missions(encodings=["ascii"], chars_min="4"), rc.missions(encodings=["koi8-r"], chars_min="10", unicode_block_filter="Cyrillic")):
        for data in (text, sxo.background(0, 32 << 20)):
            assert run_cli_product(ms, [data], radix="x", device=0) == sxo.run_cli(ms, [data], radix="x")


def test_wave_path_gives_up_and_the_other_path_takes_over(wave_forced):
    """SX_WAVE_FAIL makes the wave kernels report a wrong entry-state assumption after their count pass: stage A is finished
    in full after all (its runs were only counted) and the lane-per-region stage B produces the findings"""
    rng = random.Random(5)
    ms = rc.missions(encodings=["ascii", "utf-8"], chars_min="4")
    data = text_lines(rng, 300_000) + rng.randbytes(200_000)
    os.environ["SX_WAVE_FAIL"] = "1"
    try:
        for chunk in (None, 65536):
            assert run_cli_product(ms, [data], radix="x", device=0, chunk_bytes=chunk) == sxo.run_cli(ms, [data], radix="x")
        assert wave_windows_of_a_scan(ms, data) == 0
        # ... with the piece pipeline (kernels queued two pieces ahead): the re-scan of piece p must not share its record slot
        # with the scan of piece p + 2 — the last Mission in launch order is the one that gives up
        big = (text_lines(rng, 1 << 20) + rng.randbytes(1 << 20)) * 4
        os.environ["SX_PIECE_MIB"] = "1"
        for flags in (dict(encodings=["ascii", "utf-8"], chars_min="4"), dict(encodings=["utf-8", "ascii"], chars_min="6"),
                      dict(encodings=["koi8-r"], chars_min="5", unicode_block_filter="Cyrillic")):
            ms2 = rc.missions(**flags)
            for busiest in ("0", "1"):
                os.environ["SX_BUSIEST_LAST"] = busiest
                assert run_cli_product(ms2, [big], radix="x", device=0) == sxo.run_cli(ms2, [big], radix="x"), (flags, busiest)
    finally:
        for k in ("SX_WAVE_FAIL", "SX_PIECE_MIB", "SX_BUSIEST_LAST"):
            os.environ.pop(k, None)


from test_wave_core import DBCS_MISSIONS


@pytest.mark.parametrize("di", range(len(DBCS_MISSIONS)))
def test_wave_path_two_byte_family(wave_forced, di):
    """Big5 / Shift_JIS / EUC-KR through the wave kernels: token starts composed along the wavefront, pair codes in LDS, buffers that
    begin and end inside a token (16 KiB chunks), next to a UTF-8 Mission"""
    from test_dbcs import soup as dbcs_soup, TEXT, CODEC
    enc, flags = DBCS_MISSIONS[di]   # (EUC-JP too: tokens of up to three bytes, marks handed from lane to lane)
    ms = rc.missions(**dict(flags, encodings=flags["encodings"] + ["utf-8"]))
    rng = random.Random(4000 + di)
    txt = TEXT[enc].encode(CODEC[enc], "ignore")
    datas = [("soup", dbcs_soup(enc, rng, 300_000)), ("random", rng.randbytes(200_000)), ("text", (txt + b"\n") * (100_000 // (len(txt) + 1))),
             ("text no ascii", txt.replace(b" ", b"").replace(b"\n", b"") * 60), ("lead bytes", b"\xa4" * 9001 + b"A" + b"\xa4\xa4" * 5000 + b"\x00" * 300),
             ("three-byte tokens", (b"\x8f\xb0\xa1\x8f\xb0\xa2\x8e\xb1\x8f\xa1\x41\x8f\x41\xa4\xa2" * 9 + b"\n") * 2000)]
    for name, data in datas:
        want = sxo.run_cli(ms, [data], radix="x")
        for chunk, batches in ((None, None), (16384, "1"), (8192, "2")):
            if batches:
                os.environ["SX_WAVE_BATCHES"] = batches
            else:
                os.environ.pop("SX_WAVE_BATCHES", None)
            assert run_cli_product(ms, [data], radix="x", device=0, chunk_bytes=chunk) == want, (enc, name, chunk)
    assert wave_windows_of_a_scan(ms[:1], datas[0][1]) > 0


def test_giant_runs_take_the_wave_path():
    """a fill of accepted bytes (megabytes of spaces, '0', 0xFF in a Latin code page, 0xF6F6 in Big5) is ONE run: few runs, but every
    tile of it on the scan kernel's general path — the buffer counts as dense and is replayed a lane per window, not handed to the
    host as one region; next to a second Mission, chunked and in one piece.  The two-byte family's wave kernels take the token grid inside such
    a fill from the wavefront in front (parity of the distance: every token there has two bytes)."""
    rng = random.Random(21)
    cases = [(dict(encodings=["utf-8", "utf-16le"], chars_min="4"), b"\xff" + b" " * (6 << 20) + b"\xc3"),
             (dict(encodings=["utf-8"], chars_min="10"), rng.randbytes(5000) + b"\x80" + b"0" * (5 << 20) + rng.randbytes(3000)),
             (dict(encodings=["windows-1252", "ascii"], chars_min="4", unicode_block_filter="Latin"), b"\x00" + b"\xff" * (4 << 20) + b"\x00abcd"),
             (dict(encodings=["big5"], chars_min="4", unicode_block_filter="Cjk"), rng.randbytes(3000) + b"\x80" + b"\xf6" * ((4 << 20) + 1) + b"\n"),
             (dict(encodings=["big5"], chars_min="10", unicode_block_filter="Cjk"), rng.randbytes(3001) + b"A" + b"\xa4" * (3 << 20) + b"\xa4\x40" * 70000 + b"\n" + b"\xf6" * (1 << 20)),
             (dict(encodings=["euc-kr"], chars_min="6", unicode_block_filter="Hangul"), b"\xc7\xd1" * 40 + b"\n" + b"\xc7\xd1" * (2 << 20) + b"\n")]
    for flags, data in cases:
        ms = rc.missions(**flags)
        want = sxo.run_cli(ms, [data], radix="x")
        for chunk in (None, 1 << 20):
            assert run_cli_product(ms, [data], radix="x", device=0, chunk_bytes=chunk) == want, (flags, chunk)
        assert wave_windows_of_a_scan(ms[:1], data) > 0, flags   # (round 4: the two-byte family too — the token grid inside the fill by parity)


def test_dense_results_travel_as_16_byte_records():
    """a single Mission's wave slabs and the device-side merger of several Missions store sx_finding16 (half the bytes over PCIe):
    the packed view, its expansion through sx_result_segment (slice_index, input_file_id and the flags restored), the printed
    text, and SX_PACKED=0 must all say the same"""
    rng = random.Random(31)
    data = rng.randbytes(3 << 20) + text_lines(rng, 1 << 20)
    cases = [rc.missions(encodings=["ascii"], chars_min="4", counter_offset="1000"),
             rc.missions(encodings=["ascii", "koi8-r,,,Cyrillic", "utf-8"], chars_min="4")]
    for ms in cases:
        want = sxo.run_cli(ms, [data], radix="x")
        os.environ["SX_DEFER_MIN_BYTES"] = "1"
        try:
            sc = sx.Scanner(ms, device=0)
            for chunk in (len(data), 1 << 20):
                sc.reset()
                recs = []
                n_packed = 0
                for off in range(0, len(data), chunk):
                    res = sc.scan(data[off:off + chunk], file_id=2)
                    expanded = res.findings()
                    at = 0
                    for packed, v, n, arena, info in res.packed_segments():
                        n_packed += packed
                        for i in range(n):
                            e = expanded[at + i]
                            if packed:
                                f = v[i]
                                assert (f.position, f.mission_id, sx.PRECISION[f.flags & 3], bool(f.flags & 4)) == (e["position"], e["mission_id"], e["precision"], e["completes"])
                                assert arena[f.str_off:f.str_off + f.str_len].decode() == e["s"]
                                assert e["file_id"] == info.input_file_id == 2
                                assert e["slice_index"] == info.slice_base + (f.position - info.position0[f.mission_id]) // 4096
                            if len(ms) == 1:   # (every sx_scan call counts its slices from 0)
                                assert e["slice_index"] == (e["position"] - ms[0]["counter_offset"] - off) // 4096
                        at += n
                    recs += expanded
                    res.free()
                assert n_packed > 0, "no packed segment"
            sc.close()
            assert run_cli_product(ms, [data], radix="x", device=0) == want
            os.environ["SX_PACKED"] = "0"
            assert run_cli_product(ms, [data], radix="x", device=0) == want
        finally:
            os.environ.pop("SX_DEFER_MIN_BYTES", None); os.environ.pop("SX_PACKED", None)


def test_chunks_that_begin_inside_a_token(wave_forced):
    """two-byte family / EUC-JP: every chunk starts with the trail byte of a token begun in the chunk before (the carried decoder holds
    the lead byte) and its first kilobyte holds no byte outside the lead range — the wavefronts' way back to a token boundary then ends
    at the buffer's byte 0, where only the entry parameter says how the grid lies (found by the GPU fuzz in round 4: the look-back tile
    that begins in front of the buffer did not take it)"""
    from test_dbcs import TEXT, CODEC
    rng = random.Random(41)
    for enc, flt in (("big5", "Asian"), ("euc-kr", "Hangul"), ("euc-jp", "Asian")):
        ms = rc.missions(encodings=[enc, "ascii"], chars_min="5", output_line_len="30", unicode_block_filter=flt)
        txt = TEXT[enc].encode(CODEC[enc], "ignore").replace(b" ", b"").replace(b"\n", b"")
        pairs = bytes(b for b in txt if b >= 0x80)
        pairs = pairs[:len(pairs) // 2 * 2]
        body = b""
        while len(body) < 200_000:
            body += pairs * 3 + rng.choice([b"", b"\xa1\x40", b"xy z", b"\n"])
        data = b"A" + body      # odd offset: the pairs straddle every even chunk boundary
        want = sxo.run_cli(ms, [data], radix="x")
        for chunk in (4096, 8192, 16384):
            assert run_cli_product(ms, [data], radix="x", device=0, chunk_bytes=chunk) == want, (enc, chunk)


from test_wave_core import UTF16_MISSIONS, utf16_soup


@pytest.mark.parametrize("ui", range(len(UTF16_MISSIONS)))
def test_wave_path_utf16(wave_forced, ui):
    """UTF-16LE / BE through the wave kernels (round 4): units on the buffer's even offsets; surrogates alone, in pairs and across window and
    chunk ends; the slow mode behind a pending high surrogate.  Where the masks cannot say what the decoder does (a character kept for the
    next call at a window's end, seven high surrogates in a row) the wavefronts give the buffer back — the result is the oracle's either way"""
    flags = UTF16_MISSIONS[ui]
    be = flags["encodings"][0].endswith("be")
    codec = "utf-16-be" if be else "utf-16-le"
    ms = rc.missions(**dict(flags, encodings=flags["encodings"] + ["utf-8"]))
    rng = random.Random(8000 + ui)
    text = text_lines(rng, 150_000).decode("latin-1").encode(codec)
    datas = [("text", text), ("soup", utf16_soup(rng, 150_000, be)), ("random", rng.randbytes(200_000)),
             ("astral", ("a\U0001F600b\U00020000\U0001F601cd 中" * 9000).encode(codec)),
             ("high surrogates", utf16_soup(rng, 100_000, be, (40, 10, 14, 10, 10, 4))),
             ("odd length", text[:100_001]), ("odd start", b"x" + text[:120_000])]
    for name, data in datas:
        want = sxo.run_cli(ms, [data], radix="x")
        for chunk, batches in ((None, None), (16384, "1"), (8192, "2")):
            if batches:
                os.environ["SX_WAVE_BATCHES"] = batches
            else:
                os.environ.pop("SX_WAVE_BATCHES", None)
            assert run_cli_product(ms, [data], radix="x", device=0, chunk_bytes=chunk) == want, (flags, name, chunk)
    # a file of odd length: the next file's units begin on odd offsets (the decoder carries half a unit) — not the wave path's case
    files = [text[:100_001], text, text[:50_000]]
    assert run_cli_product(ms, files, radix="x", device=0, chunk_bytes=16384) == sxo.run_cli(ms, files, radix="x")
    assert wave_windows_of_a_scan(ms[:1], text) > 0          # the wave kernels did run
    assert wave_windows_of_a_scan(ms[:1], datas[3][1]) > 0


def test_utf16_text_is_dense_by_default():
    """UTF-16 text takes the wave path without being told to (a run per 960 bytes or more)"""
    rng = random.Random(5)
    ms = rc.missions(encodings=["utf-16le"], chars_min="4")
    data = text_lines(rng, 2_000_000).decode("latin-1").encode("utf-16-le")
    want = sxo.run_cli(ms, [data], radix="x")
    assert run_cli_product(ms, [data], radix="x", device=0, chunk_bytes=1 << 20) == want
    sc = sx.Scanner(ms, device=0)
    try:
        for off in range(0, len(data), 1 << 20):
            sc.scan(data[off:off + (1 << 20)], file_id=1).free()
        assert sc.stats().wave_windows > 0
    finally:
        sc.close()


@pytest.mark.parametrize("in_kernels", ["0", "1"])
def test_same_unicode_block_on_buffers_where_it_cannot_matter(wave_forced, in_kernels, monkeypatch):
    """-r on a UTF-8 Mission (helper.rs:279-296).  SX_WAVE_SAME=0 (and, always, Missions with -g as well): the wave kernels do not know it and
    take a buffer only if at most one lead byte that passes the filter occurs in it (the leftover carried in included) — ASCII text, text with
    one kind of multi-byte characters.  Two kinds: the wavefronts give the buffer back.  Round 5, the default: the kernels apply -r themselves
    (sx_wave_core.hpp wv_stretch_same) and take every buffer.  The result is the oracle's either way; chunks that change the kind at their boundary"""
    monkeypatch.setenv("SX_WAVE_SAME", in_kernels)
    rng = random.Random(99)
    ms = rc.missions(encodings=["utf-8"], chars_min="4", same_unicode_block=True, unicode_block_filter="All")
    ascii_text = text_lines(rng, 300_000)
    def words(alphabet, n):
        out = []
        while sum(len(w) + 1 for w in out) < n:
            out.append("".join(rng.choice(alphabet) for _ in range(rng.randrange(2, 30))))
        return (" ".join(out)).encode("utf-8")
    latin = words("abcdefg éüöàß", 200_000)            # lead byte C3 only
    cyr = words("абвгдежзийклмнопрстуфх xyz", 200_000)  # D0 and D1
    mixed = b"".join(rng.choice([latin[i:i + 16384], cyr[i:i + 16384], ascii_text[i:i + 16384]]) for i in range(0, 160_000, 16384))
    boundary = (words("é", 16384)[:16384 - 3] + "éé".encode()[:3] + words("ж", 16384)) * 4      # the kind changes where the chunks do
    for name, data in (("ascii", ascii_text), ("latin", latin), ("cyrillic", cyr), ("mixed", mixed), ("boundary", boundary), ("random", rng.randbytes(150_000))):
        want = sxo.run_cli(ms, [data], radix="x")
        for chunk in (None, 16384, 65536):
            assert run_cli_product(ms, [data], radix="x", device=0, chunk_bytes=chunk) == want, (name, chunk)
    assert wave_windows_of_a_scan(ms, ascii_text) > 0 and wave_windows_of_a_scan(ms, latin) > 0    # the wave kernels took these
    assert (wave_windows_of_a_scan(ms, cyr) > 0) == (in_kernels == "1")                            # ... and gave this one back / took it too
    # the same for UTF-16: the lead bytes of the units' UTF-8 forms
    ms16 = rc.missions(encodings=["utf-16le"], chars_min="4", same_unicode_block=True, unicode_block_filter="All")
    to16 = lambda b: b.decode("utf-8", "ignore").encode("utf-16-le")
    for name, data in (("ascii", to16(ascii_text[:150_000])), ("latin", to16(latin)), ("cyrillic", to16(cyr)), ("mixed", to16(mixed)),
                       ("astral", ("ab\U0001F600cd \U0001F601\n" * 9000).encode("utf-16-le")), ("random", rng.randbytes(150_000))):
        want = sxo.run_cli(ms16, [data], radix="x")
        for chunk in (None, 16384):
            assert run_cli_product(ms16, [data], radix="x", device=0, chunk_bytes=chunk) == want, ("utf-16le", name, chunk)
    assert wave_windows_of_a_scan(ms16, to16(ascii_text[:150_000])) > 0 and wave_windows_of_a_scan(ms16, to16(latin)) > 0
    assert (wave_windows_of_a_scan(ms16, to16(cyr)) > 0) == (in_kernels == "1")


def test_same_unicode_block_in_the_wave_kernels(wave_forced):
    """-r applied by the wave kernels (round 5): the Missions and inputs of tests/test_wave_core.py's host run of the same code, through the
    library — text that changes script every few characters, whole and in chunks"""
    import test_wave_core as twc
    rng = random.Random(404)
    for kw in twc.SAME_MISSIONS + twc.SAME_UTF16 + twc.SAME_GREP_MISSIONS:   # (the last: with -g as well)
        ms = rc.missions(**kw)
        codec = kw["encodings"][0]
        enc = lambda t: t.encode(codec, errors="replace" if not codec.startswith("utf-") else "strict")
        datas = [("scripts", enc(twc.same_text(rng, 150_000))), ("long runs", enc(twc.same_text(rng, 60_000, runs=(1, 7, 30, 64, 65, 130)))),
                 ("russian", enc(twc.russian(rng, 100_000))), ("random", rng.randbytes(100_000))]
        for name, data in datas:
            want = sxo.run_cli(ms, [data], radix="x")
            for chunk in (None, 16384):
                assert run_cli_product(ms, [data], radix="x", device=0, chunk_bytes=chunk) == want, (kw, name, chunk)
        assert wave_windows_of_a_scan(ms, datas[0][1]) > 0, kw


# ---- round 5: -g on the wave path (sx_wave_core.hpp WvWin::GC, the repairs of sx_wave.cpp) ----
def _stats_of_a_scan(ms, files):
    sc = sx.Scanner(ms, device=0)
    try:
        for i, data in enumerate(files):
            res = sc.scan(data, file_id=i + 1)
            res.free()
        return sc.stats()
    finally:
        sc.close()


def test_the_references_grep_goldens_come_out_of_the_wave_kernels(wave_forced):
    """tests/functional/run-tests:11-29 — `-q 16 -g 63` on input1, `-n 10 -q 32 -g 58` on input1 + input2, three encodings each — byte for
    byte, with every Mission on the wave kernels (sx_stats::wave_windows counts all their windows)"""
    import os as _os
    gold = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "golden")
    rd = lambda n: open(_os.path.join(gold, n), "rb").read()
    in1, in2 = rd("input1"), rd("input2")
    cases = [(dict(encodings=["UTF-8", "utf-16le", "utf-16be"], output_line_len="16", grep_char="63", ascii_filter="All-Ctrl", unicode_block_filter="Common"),
              [in1], rd("expected_output1")),
             (dict(encodings=["UTF-8", "utf-16le", "utf-16be"], chars_min="10", output_line_len="32", grep_char="58", ascii_filter="All-Ctrl",
                   unicode_block_filter="Common"), [in1, in2], rd("expected_output2"))]
    for flags, files, want in cases:
        ms = sx.missions_from_flags(**flags)
        assert ms == rc.missions(**flags)
        got = run_cli_product(ms, files, radix="x", device=0)
        assert got == want
        assert got == sxo.run_cli(ms, files, radix="x")
        st = _stats_of_a_scan(ms, files)
        windows = sum((len(f) // 4096) * (4096 // (2 * ms[0]["output_line_char_nb_max"])) for f in files)
        assert st.wave_windows >= 2 * windows, (st.wave_windows, windows)   # at least two of the three Missions replayed every window there
                                                                              # (input2's UTF-16 Missions may give a buffer back: a kept character at a window's end)


GREP_GPU = [dict(encodings=["ascii"], chars_min="4", grep_char="47"), dict(encodings=["utf-8"], chars_min="10", output_line_len="32", grep_char="58"),
            dict(encodings=["utf-8"], output_line_len="16", grep_char="63", unicode_block_filter="All"),
            dict(encodings=["utf-16le"], chars_min="5", output_line_len="16", grep_char="101"), dict(encodings=["utf-16be"], chars_min="4", grep_char="32"),
            dict(encodings=["koi8-r"], chars_min="5", unicode_block_filter="Cyrillic", grep_char="32"),
            dict(encodings=["big5"], chars_min="3", output_line_len="8", unicode_block_filter="Asian", grep_char="32"),
            dict(encodings=["euc-jp"], chars_min="3", output_line_len="8", unicode_block_filter="Cjk", grep_char="65"),
            dict(encodings=["ascii"], chars_min="4", grep_char="10")]


@pytest.mark.parametrize("gi", range(len(GREP_GPU)))
def test_wave_path_with_a_grep_char_equals_the_oracle(wave_forced, gi):
    from test_wave_core import grep_text, utf16_soup
    flags = GREP_GPU[gi]
    ms = rc.missions(**flags)
    g = ms[0]["grep_char"]
    enc = flags["encodings"][0]
    rng = random.Random(4000 + gi)
    datas = [("grep text", grep_text(rng, 600_000, g)), ("text", text_lines(rng, 300_000)), ("long lines", text_lines(rng, 200_000, 100, 900)),
             ("random", rng.randbytes(300_000)), ("no grep", bytes(c for c in text_lines(rng, 200_000, 200, 2000) if c != g))]
    if enc.startswith("utf-16"):
        codec = "utf-16-be" if enc.endswith("be") else "utf-16-le"
        datas = [(n, d if n == "random" else d.decode("latin-1").encode(codec)) for n, d in datas] + [("soup", utf16_soup(rng, 100_000, enc.endswith("be")))]
    repairs = 0
    for name, data in datas:
        want = sxo.run_cli(ms, [data], radix="x")
        for chunk, batches in ((None, None), (65536, "1"), (16384, "2")):
            if batches:
                os.environ["SX_WAVE_BATCHES"] = batches
            else:
                os.environ.pop("SX_WAVE_BATCHES", None)
            assert run_cli_product(ms, [data], radix="x", device=0, chunk_bytes=chunk) == want, (name, chunk, batches)
        os.environ["SX_WAVE_BATCHES"] = "1"
        st = _stats_of_a_scan(ms, [data])
        os.environ.pop("SX_WAVE_BATCHES", None)
        repairs += st.wave_repairs
        if name in ("grep text", "text", "no grep") and not enc.startswith(("big5", "euc")):
            assert st.wave_windows > 0, name
    if gi in (1, 2, 3):   # lines longer than a few windows without the grep char: wavefronts that begin inside one are repaired
        assert repairs > 0


def test_wave_repairs_can_be_switched_off_and_the_other_path_takes_over(wave_forced, monkeypatch):
    from test_wave_core import grep_text
    ms = rc.missions(encodings=["utf-8"], output_line_len="16", grep_char="63")
    rng = random.Random(5)
    data = bytes(c for c in text_lines(rng, 400_000, 300, 3000) if c != 63) + grep_text(rng, 100_000, 63)
    want = sxo.run_cli(ms, [data], radix="x")
    monkeypatch.setenv("SX_WAVE_BATCHES", "1")
    assert run_cli_product(ms, [data], radix="x", device=0) == want
    assert _stats_of_a_scan(ms, [data]).wave_repairs > 0
    monkeypatch.setenv("SX_WAVE_REPAIR", "0")
    assert run_cli_product(ms, [data], radix="x", device=0) == want
    assert _stats_of_a_scan(ms, [data]).wave_repairs == 0
    monkeypatch.setenv("SX_WAVE_REPAIR", "1")   # one repair launch only: a chain of wrong wavefronts is longer than that -> the other path
    assert run_cli_product(ms, [data], radix="x", device=0) == want
    monkeypatch.setenv("SX_WAVE_REPAIR", "48")
    monkeypatch.setenv("SX_WAVE_DESC", "0")     # the window-parallel writer after repairs: from the states the count pass left
    assert run_cli_product(ms, [data], radix="x", device=0) == want


# ---- round 5: SX_OPT_RESULT_ON_DEVICE ----
def test_a_dense_result_can_stay_on_the_device(wave_forced):
    """SX_OPT_RESULT_ON_DEVICE (include/stringsext_amd.h): one Mission, a string-dense buffer — the records (sx_finding16) and strings stay in
    HBM where the writer put them; sx_result_segment_device hands out the pointers; the bytes there are the bytes the host result holds;
    the host accessors fetch on first use; after the next scan the pointers are gone (SX_E_STATE)."""
    rng = random.Random(77)
    data = text_lines(rng, 3_000_000)
    for kw in (dict(encodings=["ascii"], chars_min="4"), dict(encodings=["utf-8"], chars_min="10", grep_char="58"),
               dict(encodings=["utf-16le"], chars_min="4")):
        ms = rc.missions(**kw)
        d = data.decode("latin-1").encode("utf-16-le") if kw["encodings"][0].startswith("utf-16") else data
        want = sxo.run_cli(ms, [d], radix="x")
        ref = sx.Scanner(ms, device=0)
        host = ref.scan(d, file_id=1)
        host_segs = host.packed_segments()
        sc = sx.Scanner(ms, device=0, result_on_device=True)
        try:
            res = sc.scan(d, file_id=1)
            dsegs = res.device_segments()
            assert len(dsegs) == 1 and dsegs[0][0] is not None and dsegs[0][4], kw          # one segment, in HBM, packed records
            fp, n, ap, alen, packed, info = dsegs[0]
            assert n == len(host) and alen == sum(len(a) for _, _, _, a, _ in host_segs), kw
            recs = sc.download(fp, n * 16)
            arena = sc.download(ap, alen)
            # the same findings as the host result's segments, joined (their str_off spaces are per segment: compare position + string)
            def strings(pk, records, count, ar):
                return [(records[i].position, ar[records[i].str_off:records[i].str_off + records[i].str_len], records[i].flags) for i in range(count)]
            want_list = [x for pk, r16, cnt, ar, _ in host_segs for x in strings(pk, r16, cnt, ar)]
            import ctypes as C
            r16 = (sx.Finding16 * n).from_buffer_copy(recs)
            assert strings(True, r16, n, arena) == want_list, kw
            # the host accessors fetch the segment on first use
            assert res.findings() == host.findings(), kw
            assert res.printed(n_inputs=1, radix="x") == host.printed(n_inputs=1, radix="x") and res.printed(n_inputs=1, radix="x") in want, kw
            assert res.device_segments()[0][0] is None                                       # ... and then it lies in host memory
            # a second result left on the device is gone after the next scan
            res2 = sc.scan(d, file_id=1)
            assert res2.device_segments()[0][0] is not None
            res3 = sc.scan(d[:4096 * 10], file_id=1)
            with pytest.raises(sx.SxError):
                res2.device_segments()
            with pytest.raises(sx.SxError):
                res2.segments()
            for r in (res, res2, res3):
                r.free()
        finally:
            sc.close(); host.free(); ref.close()


def test_a_sparse_result_of_one_mission_can_stay_on_the_device():
    """SX_OPT_RESULT_ON_DEVICE on the lane-per-region path (a single Mission's sparse buffer: sx_finding records, 32 bytes each): the same
    contract — device pointers, the host accessors fetch, the bytes equal the host result's"""
    rng = random.Random(78)
    from test_host_logic import synth
    data = synth(rng, 8_000_000, 1 / 400)
    ms = rc.missions(encodings=["utf-8"], chars_min="10")
    ref = sx.Scanner(ms, device=0, device_replay=True)
    host = ref.scan(data, file_id=1)
    sc = sx.Scanner(ms, device=0, device_replay=True, result_on_device=True)
    try:
        res = sc.scan(data, file_id=1)
        dsegs = res.device_segments()
        assert len(dsegs) == 1 and dsegs[0][0] is not None and not dsegs[0][4]       # one segment, in HBM, sx_finding records
        fp, n, ap, alen, packed, info = dsegs[0]
        assert n == len(host) and n > 1000
        fb, ab = host.raw()
        assert sc.download(fp, n * 32) == fb and sc.download(ap, alen) == ab
        assert res.findings() == host.findings()
        assert res.device_segments()[0][0] is None
        res.free()
    finally:
        sc.close(); host.free(); ref.close()


def test_a_result_left_on_the_device_dies_with_the_next_buffer_and_with_the_context(tmp_path):
    """(1) every BUFFER reuses the context's device block — the chunks of one sx_scan_file call too: a chunk's result that was
    left in HBM answers SX_E_STATE once the next chunk has been scanned (before: the epoch advanced per API call only and it handed out the next
    chunk's bytes); (2) sx_destroy frees the block: a result that outlives its context answers SX_E_STATE instead of reading freed memory."""
    rng = random.Random(79)
    data = text_lines(rng, 3 * (1 << 20) + 12345)
    ms = rc.missions(encodings=["ascii"], chars_min="4")
    path = tmp_path / "in.bin"
    path.write_bytes(data)
    ref = sx.Scanner(ms, device=0)
    want = [r for r in ref.scan_file(str(path), chunk_bytes=1 << 20)]
    sc = sx.Scanner(ms, device=0, result_on_device=True)
    got = sc.scan_file(str(path), chunk_bytes=1 << 20)
    assert len(got) == len(want) >= 3
    # the last chunk's result is the one that may still lie in HBM; it is the host result's
    assert got[-1].findings() == want[-1].findings()
    for r, w in zip(got[:-1], want[:-1]):
        try:
            segs = r.device_segments()
        except sx.SxError:
            continue                                   # it was left in HBM and a later chunk has reused the block: refused, not wrong
        assert all(s[0] is None for s in segs)          # ... or it was moved to the host in time: then it is right
        assert r.findings() == w.findings()
    last = sc.scan(data[:1 << 20], file_id=1)
    assert last.device_segments()[0][0] is not None
    sc.close()
    with pytest.raises(sx.SxError):
        last.device_segments()
    with pytest.raises(sx.SxError):
        last.segments()
    for r in want:
        r.free()
    ref.close()


def test_eucjp_fills_stay_on_the_wave_path(wave_forced):
    """a fill of lead-range bytes longer than 64 KiB made EUC-JP's wavefronts give the buffer back (token lengths differ: no
    parity).  Without an 8E / 8F among the bytes walked over every token has two bytes there, and the hang-over follows from the wavefront in
    front as it does for Big5 (WaveParams::wave_grid, two bits); with one, the wavefronts still give up — and tell the ones behind them.  The
    result is the oracle's either way."""
    from test_dbcs import TEXT
    txt = TEXT["euc-jp"].encode("euc_jp", "ignore")
    ms = rc.missions(encodings=["euc-jp"], chars_min="4", unicode_block_filter="Asian")   # (a Mission the wave path takes: UTF-8 forms of one length)
    for fill in (0xA4, 0xF6, 0xFE):
        for odd in (0, 1):
            f = bytes([fill])
            data = txt * 150 + f * (300_000 + odd) + b"A" + f * (150_001 + odd) + txt * 150 + f * (70_000 + odd) + txt * 20
            want = sxo.run_cli(ms, [data], radix="x")
            for chunk in (None, 65536):
                assert run_cli_product(ms, [data], radix="x", device=0, chunk_bytes=chunk) == want, (fill, odd, chunk)
            assert wave_windows_of_a_scan(ms, data) > 0, (fill, odd)       # the wave kernels kept the buffer
    # 8E / 8F inside a long fill: tokens of two and three bytes — no parity; given back, the other path's result
    data = txt * 100 + (b"\xa4" * 999 + b"\x8f") * 300 + b"\n" + (b"\xb0" * 1001 + b"\x8e") * 200 + txt * 100
    want = sxo.run_cli(ms, [data], radix="x")
    for chunk in (None, 16384):
        assert run_cli_product(ms, [data], radix="x", device=0, chunk_bytes=chunk) == want, chunk
