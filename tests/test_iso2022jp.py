"""ISO-2022-JP (help.rs:60; Encoding::for_label at mission.rs:681): the last encoding of the reference's list.  Escape sequences
select the character set, so the decoder's state at a byte is not derivable from the bytes near it: a Mission with it is ONE
sequential pass of FindingCollection::from on the host (csrc/sx_stage_b.cpp host_sequential_mission), next to the device's work
for the other Missions.  Oracle and product decoders are written separately (oracle/sxo.c dec_iso2022jp: the WHATWG states by
name; csrc/sx_codec_core.hpp ddec_iso2022jp: selected set x position in a token); the hand-derived vectors are in
tests/golden/decoder_vectors.py."""
import random

import pytest

import refconfig as rc
import stringsext_amd as sx
import sxo_binding as sxo
from product_harness import run_cli_product
from test_decoder_vectors import OracleDecoder, ProductDecoder

TXT = "日本語のテキスト、ｶﾀｶﾅ、ASCII text 123 ¥‾ 漢字かな交じり文。"
ESCS = [b"\x1b(B", b"\x1b(J", b"\x1b(I", b"\x1b$@", b"\x1b$B"]


def soup(rng, n):
    out = bytearray()
    while len(out) < n:
        r = rng.random()
        if r < 0.35: out += TXT[rng.randrange(len(TXT)):][:rng.randrange(1, 40)].encode("iso2022_jp_ext", "ignore")
        elif r < 0.5: out += rng.choice(ESCS) + bytes(rng.randrange(0x21, 0x7F) for _ in range(rng.randrange(0, 60)))
        elif r < 0.6: out += rng.randbytes(rng.randrange(1, 60))
        elif r < 0.7: out += b"plain ascii text %d " % rng.randrange(1000)
        elif r < 0.8: out += bytes(rng.choice([0x1B, 0x24, 0x28, 0x42, 0x4A, 0x49, 0x40, 0x0E, 0x0F, 0x80, 0x5C, 0x7E, 0x21, 0x41]) for _ in range(rng.randrange(1, 30)))
        elif r < 0.85: out += b"\x00" * rng.randrange(1, 300)
        else: out += rng.choice(ESCS) + rng.choice(ESCS)
    return bytes(out[:n])


def decode_all(dec, data, piece):
    text = b""
    for off in range(0, len(data), piece):
        rest = data[off:off + piece]
        for _ in range(10_000):
            r, rd, wr, got = dec.step(rest, False)
            text += got
            rest = rest[rd:]
            if r == "E":   # (M: the loop calls again; F: the harness' 256-byte output buffer was full)
                break
    return text


def test_oracle_and_product_decoders_agree_and_match_cpython_on_clean_text():
    rng = random.Random(1)
    clean = (TXT * 20).encode("iso2022_jp_ext")
    for which in (OracleDecoder, ProductDecoder):
        assert decode_all(which(71), clean, 1 << 20).decode("utf-8") == clean.decode("iso2022_jp_ext")   # CPython: an independent decoder
    for seed in range(30):
        data = soup(random.Random(seed), 20_000)
        for piece in (1 << 20, 128, 7, 1):
            assert decode_all(OracleDecoder(71), data, piece) == decode_all(ProductDecoder(71), data, piece), (seed, piece)
    del rng


FLAGS = [dict(chars_min="4", unicode_block_filter="All"), dict(chars_min="3", output_line_len="16", unicode_block_filter="Cjk"),
         dict(chars_min="2", unicode_block_filter="Asian", same_unicode_block=True), dict(chars_min="5", grep_char="0x20", unicode_block_filter="All"),
         dict(chars_min="10", unicode_block_filter="Kana"), dict(chars_min="0", unicode_block_filter="All"), dict(chars_min="4")]


@pytest.mark.parametrize("flags", FLAGS, ids=lambda f: "n" + f["chars_min"] + "-" + f.get("unicode_block_filter", "default"))
def test_host_pass_equals_oracle(flags):
    rng = random.Random(len(repr(flags)))
    data = soup(rng, 120_000)
    ms = rc.missions(encodings=["iso-2022-jp", "utf-8"], **flags)
    assert ms[0]["encoding"] == 71 and sx.encoding_name(71) == "ISO-2022-JP" and sx.encoding_for_label("csiso2022jp") == 71
    want = sxo.run_cli(ms, [data], radix="x")
    assert len(want) > 200 and b"(a ISO-2022-JP)" in want
    assert run_cli_product(ms, [data], radix="x") == want
    for chunk in (4096, 8192, 65536):
        assert run_cli_product(ms, [data], radix="x", chunk_bytes=chunk, flush_at_eof=True) == sxo.run_cli(ms, [data], radix="x", flush_at_eof=True), chunk
    files = [data[:50_001], b"", data[50_001:]]
    assert run_cli_product(ms, files, radix="x") == sxo.run_cli(ms, files, radix="x")


@pytest.mark.gpu
def test_iso2022jp_next_to_device_missions_on_the_gpu():
    rng = random.Random(5)
    data = soup(rng, 400_000) + rng.randbytes(300_000) + soup(rng, 100_000)
    ms = rc.missions(encodings=["utf-8", "iso-2022-jp", "utf-16le", "ascii"], chars_min="5", unicode_block_filter="All")
    want = sxo.run_cli(ms, [data], radix="x")
    for chunk in (None, 65536):
        assert run_cli_product(ms, [data], radix="x", device=0, chunk_bytes=chunk) == want, chunk
    # device-resident input: the Mission's bytes come to the host in pieces
    sc = sx.Scanner(ms, device=0)
    d = sc.alloc(len(data)); sc.upload(d, data)
    res = sc.scan_device(d, len(data), file_id=1)
    assert sx.OUTPUT_BOM + res.printed(n_inputs=1, radix="x") + b"\n" == want
    res.free(); sc.free(d); sc.close()
    # no stage A for it, no shards
    sc = sx.Scanner(rc.missions(encodings=["iso-2022-jp"]), device=0)
    d = sc.alloc(8192); sc.upload(d, bytes(8192))
    with pytest.raises(sx.SxError):
        sc.device_runs(0, d, 8192, stream_parity=0, min_chars=4)
    sc.free(d); sc.close()
