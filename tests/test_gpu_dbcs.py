"""GPU parity for the double-byte encodings (Big5, EUC-JP; BASELINE config 5): stage A's token classifier
(csrc/sx_kernels.hip scan_kernel_dbcs) against the oracle's sequential decoder, and the whole path —
kernels, device and host replay — against the oracle's full scan.  Through the C-ABI."""
import random
import zlib

import pytest

import refconfig as rc
import sxo_binding as sxo
from product_harness import run_cli_product
from test_dbcs import ALL, CODEC, DBCS_FLAGS, ENCS, TEXT, soup, text_lines
from test_gpu_parity import device_runs

pytestmark = pytest.mark.gpu

RUNS = {
    "all": dict(chars_min="4", unicode_block_filter=ALL),
    "cjk": dict(chars_min="3", unicode_block_filter="Cjk"),
    "asian_n10": dict(chars_min="10", unicode_block_filter="Asian"),
    "kana_noascii": dict(chars_min="2", unicode_block_filter="Kana", ascii_filter="None"),
    "common": dict(chars_min="5", unicode_block_filter="Common"),
    "odd_af": dict(chars_min="4", unicode_block_filter="Cjk", ascii_filter="0x7ffffffe000000007ffffffe00000000"),
}


@pytest.mark.parametrize("enc", ENCS)
@pytest.mark.parametrize("name", sorted(RUNS))
def test_device_runs_equal_oracle_runs(enc, name):
    m = rc.missions(encodings=[enc], **RUNS[name])[0]
    rng = random.Random(zlib.crc32((enc + name).encode()))
    txt = TEXT[enc].encode(CODEC[enc], "ignore")
    datas = [
        soup(enc, rng, 300_000),
        rng.randbytes(1 << 20),
        txt * 300,                                        # no byte outside the lead range for kilobytes
        (b"\x88" if enc == "shift_jis" else b"\xa4") * 5000 + b"A" + (b"\x88" if enc == "shift_jis" else b"\xa4") * 5001 + b"\n" + txt * 40,   # long stretches of one lead byte, both parities
        b"\xb1\xdf\x80" * 3000 + b"\xa0\xfd" * 100 + txt * 5,        # Shift_JIS: one-byte characters >= 0x80
        b"\x8f\xb0\xa1" * 3000 + b"\x8f" * 3001 + txt * 10 + b"\x8e\xb1" * 2000,
        txt[:1023], txt[:1025], txt[1:18], b"abcdefghijkl", b"", b"\xa4", b"\xa4\x40",
        b"A" * 5000 + rng.randbytes(3000) + txt * 7 + b"\x00" * 100 + b"zz" * 3000,
        # gb18030: rows of `lead digit lead digit ...` — every other candidate is a token, from the row's first; rows across lanes, tiles
        # and sub-chunks, at both parities, with pointers in range (81 30 .., 82 35 .., 90 30 ..: astral) and out of it (84 32 .., 8F 39 ..)
        b"\x81\x30" * 5001 + b"x" + b"\x81\x30" * 700 + b"yy" + b"\x84\x32\x81\x30" * 900 + b"\x90\x30\x81\x30\x82\x35" * 1200 + b"z" +
        b"\x8f\x39" * 333 + b"\x81\x30\x81" * 400 + b"\x82\x35\x8f\x39\x81\x30" * 800 + txt * 3,
        b"".join(rng.choice([b"\x81\x30", b"\x82\x35", b"\x84\x32", b"\x90\x30", b"\xfe\x39", b"\x84\x31", b"\xe3\x32", b"7", b"\x81", b"A"])
                 for _ in range(150_000)),
    ]
    for di, data in enumerate(datas):
        for sub in (1024, 4096, 65536):
            if sub != 65536 and len(data) > 400_000:
                continue
            got, mc = device_runs(m, data, subchunk=sub)
            want = sxo.runs(m, data, min_chars=mc)
            # (gb18030 / GBK: exact since round 4 — of a row of `lead digit lead digit` candidates every other one is a token, a character
            # if its pointer is in range, marked by its filter; rounds 2-3 reported a superset there)
            assert got == want, (enc, name, di, sub, len(got), len(want),
                                 next(((a, b) for a, b in zip(got, want) if a != b), None))


@pytest.mark.parametrize("enc", ENCS)
def test_long_fills_of_lead_range_bytes(enc):
    """Format and deleted-entry fills of disk images (0xF6, 0xE5, ...) lie inside the lead ranges: for kilobytes no byte tells where
    tokens start.  The scan kernel takes the grid from the sub-chunk in front (a published hang-over + parity) instead of walking
    to the fill's beginning: fills across many sub-chunks, both parities, chunks that begin inside a fill, EUC-JP's 8E / 8F."""
    rng = random.Random(zlib.crc32(enc.encode()) + 7)
    txt = TEXT[enc].encode(CODEC[enc], "ignore")
    m = rc.missions(encodings=[enc], chars_min="4", unicode_block_filter=ALL)[0]
    ms = rc.missions(encodings=[enc, "utf-8"], chars_min="4", unicode_block_filter=ALL)
    for fill in (0xF6, 0xE5, 0xA4, 0x81, 0xFE, 0x8F, 0x8E, 0x39):
        for odd in (0, 1):
            f = bytes([fill])
            data = (soup(enc, rng, 3000) + f * (40_000 + odd) + b"A" + f * (70_001 + odd) + txt * 5 + f * (9000 + odd) + b"\x8f" + f * 5000 +
                    b"1" + f * (12_345 + odd) + txt[:77] + b"\n")
            for sub in (1024, 4096, 65536):
                got, mc = device_runs(m, data, subchunk=sub)
                want = sxo.runs(m, data, min_chars=mc)
                assert got == want, (enc, hex(fill), odd, sub, len(got), len(want), next(((a, b) for a, b in zip(got, want) if a != b), None))
            want = sxo.run_cli(ms, [data], radix="x")
            for chunk in (None, 8192, 4096 * 7):
                assert run_cli_product(ms, [data], radix="x", device=0, chunk_bytes=chunk) == want, (enc, hex(fill), odd, chunk)


@pytest.mark.parametrize("enc", ENCS)
@pytest.mark.parametrize("flags", DBCS_FLAGS, ids=lambda f: "n" + f["chars_min"] + "-" + f["unicode_block_filter"])
def test_end_to_end_equals_oracle(enc, flags):
    rng = random.Random(zlib.crc32((enc + "gpu" + repr(sorted(flags.items()))).encode()))
    files = [soup(enc, rng, 400_000), soup(enc, rng, 20_001), b"", TEXT[enc].encode(CODEC[enc], "ignore") * 200]
    ms = rc.missions(encodings=[enc, "utf-8", "utf-16le"], **flags)
    want = sxo.run_cli(ms, files, radix="x")
    for chunk, sub, dev_replay in ((None, 0, None), (16384, 1024, None), (None, 0, True), (65536, 4096, True)):
        got = run_cli_product(ms, files, radix="x", chunk_bytes=chunk, device=0, subchunk_bytes=sub, device_replay=dev_replay)
        assert got == want, (chunk, sub, dev_replay)


@pytest.mark.parametrize("enc", ENCS)
def test_text_lines_on_the_gpu(enc):
    """Lines of CJK text, every one across several window starts: the device cuts the runs into pieces (gb18030 / GBK: those
    that verify as exact) and replays a region per window; 120 000 lines so that stage B runs on the device."""
    data = text_lines(enc, random.Random(zlib.crc32((enc + "lines").encode())), 120_000)
    for flags in (dict(chars_min="4", unicode_block_filter=ALL), dict(chars_min="2", output_line_len="12", unicode_block_filter="Cjk")):
        ms = rc.missions(encodings=[enc], **flags)
        want = sxo.run_cli(ms, [data], radix="x")
        for chunk, dev_replay in ((None, None), (1 << 20, True)):
            assert run_cli_product(ms, [data], radix="x", chunk_bytes=chunk, device=0, device_replay=dev_replay) == want, (flags, chunk)


def test_c5_six_missions_on_a_planted_image():
    """BASELINE config 5 with the per-encoding filters SURVEY 8(a) recommends, missions built by the product's
    front end from the literal flag strings."""
    import stringsext_amd as sx
    flags = dict(encodings=["utf-8,,,African", "utf-16le,,,African", "utf-16be,,,African", "big5,,,Cjk", "euc-jp,,,Asian",
                            "koi8-r,,,Cyrillic"], chars_min="10")
    ms = sx.missions_from_flags(**flags)
    assert ms == rc.missions(**flags)
    rng = random.Random(5)
    data = bytearray(sxo.background(0, 8 << 20))
    texts = [("Բարեւ աշխարհ — שלום עולם — مرحبا بالعالم /usr/share/doc", ["utf-8", "utf-16-le", "utf-16-be"]),
             (TEXT["big5"], ["big5hkscs"]), (TEXT["euc-jp"], ["euc_jp"]), ("Привет, мир! Доброе утро, страна.", ["koi8-r"])]
    pos = 3000
    while pos + 2000 < len(data):
        t, codecs = rng.choice(texts)
        b = t.encode(rng.choice(codecs), "ignore")
        data[pos:pos + len(b)] = b
        pos += rng.choice([4096 - 40, 65536 - 17, 128 * 3 + 5, 20000])
    data = bytes(data)
    want = sxo.run_cli(ms, [data], radix="x")
    assert want.count(b"(d Big5)") > 50 and want.count(b"(e EUC-JP)") > 50 and want.count(b"(f KOI8-R)") > 20
    for dev_replay in (None, True):
        assert run_cli_product(ms, [data], radix="x", device=0, device_replay=dev_replay) == want


def test_a_gb18030_row_of_megabytes_is_scanned_in_bounded_time():
    """`81 30 81 30 ...` for megabytes: every other candidate is a four-byte token, and which one is known only from the row's beginning —
    a sub-chunk's wavefront takes it from the one in front, which publishes it on entering its last tile.  The wait is bounded (ADVICE r4:
    after ~1 ms the row is marked as a superset and stage B decides); the result is the oracle's and the scan takes seconds, not minutes."""
    import time
    ms = rc.missions(encodings=["gb18030"], chars_min="4", unicode_block_filter=ALL)
    for odd in (0, 1):
        data = b"text in front\n" + b"\x81" * odd + b"\x81\x30" * (3 << 20) + b"\nand behind it " + "中文字符".encode("gb18030") * 50 + b"\n"
        want = sxo.run_cli(ms, [data], radix="x")
        t0 = time.time()
        got = run_cli_product(ms, [data], radix="x", device=0)
        dt = time.time() - t0
        assert got == want, odd
        assert dt < 60, dt


def test_regions_inside_a_fill_of_lead_range_bytes_replay_in_bounded_time(monkeypatch):
    """stage B's walk back to a token boundary (sx_replay_core.hpp dbcs_sync_before) went to the nearest byte outside the lead
    range however far — megabytes of valid two-byte characters the filter rejects, with a short accepted string every kilobyte, made every
    region walk to the fill's beginning.  It ends at a sub-chunk start now, where stage A has published the token grid.  Lane-per-region path
    forced; both parities of the fill."""
    import time
    monkeypatch.setenv("SX_WAVE_REPLAY", "0")
    ms = rc.missions(encodings=["euc-kr"], chars_min="4", unicode_block_filter="Kana")
    fill, kana = "가".encode("euc_kr"), "あいうえおかきくけこさし".encode("euc_kr")
    for odd in (0, 1):
        data = b"x" + b"\xb0" * odd + (fill * 500 + kana) * 3000 + b"\n"
        want = sxo.run_cli(ms, [data], radix="x")
        t0 = time.time()
        got = run_cli_product(ms, [data], radix="x", device=0, device_replay=True)
        dt = time.time() - t0
        assert got == want, odd
        assert dt < 30, dt


@pytest.mark.parametrize("enc", ["gb18030", "gbk"])
def test_gb18030_four_byte_passage_across_sub_chunk_starts(monkeypatch, enc):
    """a passage of four-byte characters the filter rejects (lead digit lead digit ..., nothing but lead and digit bytes) that
    crosses sub-chunk starts at every phase, an accepted string right behind it.  Stage B's walk back to a token boundary must not end at
    a sub-chunk start there: stage A publishes the hang-over of the TWO-byte grammar, which is two bytes off inside a four-byte token.
    Lane-per-region path forced; sub-chunks of 4 KiB so that many starts fall inside the passage."""
    monkeypatch.setenv("SX_WAVE_REPLAY", "0")
    ms = rc.missions(encodings=[enc], chars_min="4", unicode_block_filter="Cjk")
    four = "ᠠᠡᠢᠣᠤ".encode("gb18030")       # Mongolian: four bytes each, UTF-8 lead E1 — not in Cjk
    assert len(four) == 20
    good = "中文字符串测试内容".encode("gb18030")
    for shift in range(4):
        data = b"x" * (1 + shift) + (four * 3000 + good) * 12 + b"\n" + four * 20000 + good + b"\n"
        want = sxo.run_cli(ms, [data], radix="x")
        for sub in (4096, 0):
            got = run_cli_product(ms, [data], radix="x", device=0, device_replay=True, subchunk_bytes=sub)
            assert got == want, (enc, shift, sub)
