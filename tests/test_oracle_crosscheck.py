"""Secondary, NON-authoritative pin of the oracle's decoders (SURVEY §8c lists them as "parity
unpinned": no reference vector exists): which bytes belong to validly encoded characters must agree
with CPython's independent UTF-8 / UTF-16 decoders — the un-read rule after a bad continuation
byte, the narrowed second-byte ranges after E0 / ED / F0 / F4, C0/C1/F5..FF, surrogate pairing.
The oracle is asked through its run finder with a pass-everything filter; NUL (rejected by every
ascii filter, mission.rs:225) and — for UTF-16 — nothing else splits the stretches."""
import codecs
import itertools
import random

import pytest

import refconfig as rc
import sxo_binding as sxo

_errors = []


def _rec(e):
    _errors.append((e.start, e.end))
    return ("", e.end)


codecs.register_error("sx-record", _rec)


def python_valid_mask(data, codec, unit):
    """1 for every byte that CPython decodes as part of a character (and that is not NUL)."""
    _errors.clear()
    text = data.decode(codec, errors="sx-record")
    mask = bytearray(b"\x01" * len(data))
    for a, b in _errors:
        for i in range(a, min(b, len(data))):
            mask[i] = 0
    # a truncated tail that CPython silently keeps for a later call: not a character yet
    enc_len = len(text.encode(codec, errors="surrogatepass"))
    bad = sum(b - a for a, b in _errors)
    for i in range(enc_len + bad, len(data)):
        mask[i] = 0
    # NUL never passes the ascii filter
    pos = 0
    for ch in text:
        n = len(ch.encode(codec, errors="surrogatepass"))
        while pos < len(data) and not mask[pos]:
            pos += 1
        if ch == "\x00":
            for i in range(pos, pos + n):
                mask[i] = 0
        pos += n
    return bytes(mask)


def oracle_valid_mask(data, enc):
    m = rc.mission(encoding=enc, chars_min_nb=1, af=rc.AF_ALL, ubf=rc.UBF_ALL, output_line_char_nb_max=64)
    mask = bytearray(len(data))
    for a, b, _ in sxo.runs(m, data, stream_parity=0, min_chars=1, cap=1 << 20):
        for i in range(a, b):
            mask[i] = 1
    return bytes(mask)


def utf8_cases():
    rng = random.Random(1)
    leads = [0x41, 0x7F, 0x80, 0xBF, 0xC0, 0xC1, 0xC2, 0xDF, 0xE0, 0xE1, 0xEC, 0xED, 0xEE, 0xEF, 0xF0, 0xF1, 0xF3, 0xF4, 0xF5, 0xFF]
    seconds = [0x00, 0x41, 0x7F, 0x80, 0x8F, 0x90, 0x9F, 0xA0, 0xBF, 0xC0, 0xC2, 0xE0, 0xED, 0xF0, 0xF4]
    out = bytearray()
    for a, b in itertools.product(leads, seconds):          # every lead x second-byte class, then what follows
        for tail in (b"", b"\x80", b"\xbf\x80", b"\x80\x80\x41", b"\x41"):
            out += b"x" + bytes([a, b]) + tail + b"y\xffz"
    for _ in range(3000):                                   # soup of interesting bytes
        out += bytes(rng.choice(leads + seconds) for _ in range(rng.randrange(1, 9))) + b" ok "
    out += "valid: é € 𝔘 ퟿  \U0010ffff".encode("utf-8", errors="surrogatepass")
    return bytes(out)


def test_utf8_validity_agrees_with_cpython():
    data = utf8_cases()
    got, want = oracle_valid_mask(data, 1), python_valid_mask(data, "utf-8", 1)
    bad = [i for i in range(len(data)) if got[i] != want[i]]
    assert not bad, (bad[:5], data[max(0, bad[0] - 6):bad[0] + 6].hex())


@pytest.mark.parametrize("enc,codec", [(2, "utf-16-le"), (3, "utf-16-be")])
def test_utf16_validity_agrees_with_cpython(enc, codec):
    rng = random.Random(2)
    units = [0x0041, 0x00E9, 0x20AC, 0xD7FF, 0xD800, 0xDBFF, 0xDC00, 0xDFFF, 0xE000, 0xFFFD, 0xFFFF, 0x0001]
    out = bytearray()
    for a, b, c in itertools.product(units, repeat=3):
        for u in (0x0078, a, b, c, 0x0079):
            out += u.to_bytes(2, "little" if enc == 2 else "big")
    for _ in range(4000):
        out += rng.choice(units).to_bytes(2, "little" if enc == 2 else "big")
    data = bytes(out)
    got, want = oracle_valid_mask(data, enc), python_valid_mask(data, codec, 2)
    bad = [i for i in range(len(data)) if got[i] != want[i]]
    assert not bad, (bad[:5], data[max(0, bad[0] - 8):bad[0] + 8].hex())
