"""Stage B's fast pre-pass (round 5, csrc/sx_replay_dev.hip replay_fast_kernel): regions that are one run inside one window are
settled from the run record and a walk back to the start of the run's decoder call; everything else stays with the general
lane-per-region kernel.  The two must be indistinguishable: every case below is scanned with the pre-pass (the default), without
it (SX_FAST_REPLAY=0) and by the oracle, and the pre-pass must really have taken regions (sx_stats::fast_regions)."""
import random

import pytest

import refconfig as rc
import stringsext_amd as sx
import sxo_binding as sxo

pytestmark = pytest.mark.gpu
SEED = 0x5EED5EED5EED5EED

SHAPES = {
    "african_n10": dict(encodings=["utf-8"], chars_min="10", unicode_block_filter="African"),
    "common_n4": dict(encodings=["utf-8"], chars_min="4"),
    "n6_q8": dict(encodings=["utf-8"], chars_min="6", output_line_len="8"),          # windows of 16 bytes: every edge rule, all the time
    "n3_q6": dict(encodings=["utf-8"], chars_min="3", output_line_len="6"),
    "all_n8": dict(encodings=["utf-8"], chars_min="8", unicode_block_filter="All"),   # three- and four-byte characters pass
    "cyr_only": dict(encodings=["utf-8"], chars_min="5", unicode_block_filter="Cyrillic", ascii_filter="None"),
    "n12_q12": dict(encodings=["utf-8"], chars_min="12", output_line_len="12"),       # every run of the list has >= q characters
}


def scan(ms, data, chunk=None, fast=True, monkeypatch=None):
    if fast:
        monkeypatch.delenv("SX_FAST_REPLAY", raising=False)
    else:
        monkeypatch.setenv("SX_FAST_REPLAY", "0")
    sc = sx.Scanner(ms, device=0, device_replay=True)
    out = bytearray(sx.OUTPUT_BOM)
    step = chunk or max(len(data), 1)
    try:
        for off in range(0, len(data), step):
            res = sc.scan(data[off:off + step], file_id=1)
            out += res.printed(n_inputs=1, radix="x")
            res.free()
        st = sc.stats()
        return bytes(out) + b"\n", st.fast_regions, (st.general_regions if st.wave_windows == 0 else -1)
    finally:
        sc.close()


def soup(rng, n):
    """UTF-8 where every rule of the pre-pass is exercised: runs of every length next to window edges, valid-but-rejected bytes
    (no call boundary), every kind of malformed sequence (a call boundary in front of or behind it), characters of two to four
    bytes that straddle window starts, the narrowed second bytes of E0 / ED / F0 / F4."""
    out = bytearray()
    ascii_run = "abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789 /._-"
    wide = "éüñ" + "жЖдя" + "אבגד" + "مرحبا" + "Ա Ձ" + "€日本語한" + "😀𝄞"
    bad = [b"\x80", b"\xbf", b"\xc0", b"\xc1", b"\xf5", b"\xff", b"\xc3", b"\xe2\x82", b"\xf0\x9f\x98", b"\xe0\x80\x80", b"\xed\xa0\x80",
           b"\xf0\x80\x80\x80", b"\xf4\x90\x80\x80", b"\xe2", b"\xf0\x9f", b"\x80\x80\x80\x80\x80"]
    ctrl = [b"\x00", b"\x01", b"\x1f", b"\x7f", b"\x09", b"\x0a"]
    while len(out) < n:
        k = rng.random()
        if k < 0.30:
            out += "".join(rng.choice(ascii_run) for _ in range(rng.choice([1, 2, 3, 5, 8, 10, 11, 12, 13, 16, 24, 40, 63, 64, 65, 90]))).encode()
        elif k < 0.42:
            out += "".join(rng.choice(ascii_run + wide) for _ in range(rng.randrange(1, 30))).encode()
        elif k < 0.62:
            out += rng.choice(bad)
        elif k < 0.75:
            out += rng.choice(ctrl) * rng.randrange(1, 4)
        elif k < 0.90:
            out += bytes(rng.randrange(256) for _ in range(rng.choice([1, 3, 17, 60, 200, 700])))
        else:   # bring the next token next to a window / slice edge
            pad = (-len(out)) % rng.choice([16, 128, 128, 4096]) - rng.randrange(0, 20)
            if pad > 0:
                out += bytes(rng.choice(b"\x00\x01\xff\x80 ") for _ in range(pad))
    return bytes(out[:n])


@pytest.mark.parametrize("shape", sorted(SHAPES))
def test_prepass_general_kernel_and_oracle_agree_on_random_bytes(shape, monkeypatch):
    ms = rc.missions(**SHAPES[shape])
    data = sxo.background(0, 6 << 20, SEED)
    want = sxo.run_cli(ms, [data], radix="x")
    got, fast, general = scan(ms, data, monkeypatch=monkeypatch)
    assert got == want
    # (general == -1: string-dense for this Mission, the wave kernels replayed every window; 0 regions: no run at all, e.g. Cyrillic only)
    if shape == "n12_q12":
        assert fast == 0                                       # a run of >= q characters is never the pre-pass'
    elif shape in ("african_n10", "all_n8"):
        assert fast > 4 * general > 0, (fast, general)         # binary data, windows of 128 bytes: nearly every region is one run in one window
    got0, fast0, general0 = scan(ms, data, fast=False, monkeypatch=monkeypatch)
    assert got0 == want and fast0 == 0 and general0 <= 0


@pytest.mark.parametrize("shape", sorted(SHAPES))
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_prepass_general_kernel_and_oracle_agree_on_utf8_soup(shape, seed, monkeypatch):
    ms = rc.missions(**SHAPES[shape])
    monkeypatch.setenv("SX_WAVE_REPLAY", "0")   # (the soup is string-dense: without this the wave kernels replay every window)
    rng = random.Random(seed * 1000 + len(shape))
    data = soup(rng, 3_000_000 + rng.randrange(4096))
    want = sxo.run_cli(ms, [data], radix="x")
    for chunk in (None, 1 << 20, 64 << 10):
        got, fast, general = scan(ms, data, chunk=chunk, monkeypatch=monkeypatch)
        assert got == want, (shape, seed, chunk)
    if shape in ("african_n10", "all_n8", "n12_q12"):
        assert fast + general > 0
    got0, _, _ = scan(ms, data, fast=False, monkeypatch=monkeypatch)
    assert got0 == want


def test_prepass_next_to_slabs_pieces_and_small_regions(monkeypatch):
    """the switches that change how the run list reaches pass 1: slabs of it, no pieces, a tiny cache (regions without a slot)"""
    ms = rc.missions(**SHAPES["common_n4"])
    monkeypatch.setenv("SX_WAVE_REPLAY", "0")
    rng = random.Random(77)
    data = soup(rng, 2_500_000)
    want = sxo.run_cli(ms, [data], radix="x")
    for env in ({"SX_SLABS": "5"}, {"SX_NO_PIECES": "1"}, {"SX_REPLAY_CACHE_MIB": "1"}, {"SX_STITCH_BLOCK": "7"}, {"SX_DEVICE_JOIN_MIN": "1"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        got, fast, general = scan(ms, data, monkeypatch=monkeypatch)
        assert got == want, env
        for k in env:
            monkeypatch.delenv(k)
