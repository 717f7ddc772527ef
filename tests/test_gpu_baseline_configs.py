"""Parity on BASELINE.json's configurations, on the GPU, through the C-ABI:
C1 full text diff, C2 slice + whole-size invariants, C3(i) run records vs the oracle on a 1 GiB
prefix, C3(ii) "disk image" with planted records (full text diff on 256 MiB), C5's KOI8-R part.
The oracle is the checker; sizes are what it finishes in seconds."""
import ctypes
import random

import pytest

import refconfig as rc
import stringsext_amd as sx
import sxo_binding as sxo

pytestmark = pytest.mark.gpu
SEED = 0x5EED5EED5EED5EED


def product_missions(**flags):
    """Missions as the PRODUCT's front end builds them from the literal BASELINE flag strings
    (sx_missions_from_flags, csrc/sx_front.cpp); tests/refconfig.py only checks them."""
    ms = sx.missions_from_flags(**flags)
    assert ms == rc.missions(**flags)
    return ms


CORPUS = [
    "/usr/lib/x86_64-linux-gnu/libc.so.6", "C:\\Windows\\System32\\drivers\\etc\\hosts", "GET /index.html HTTP/1.1",
    "Բարեւ Ձեզ, ինչպես եք այսօր", "Հայաստանի Հանրապետություն", "שלום עולם, זוהי בדיקה של מחרוזות", "ירושלים של זהב",
    "مرحبا بالعالم هذا اختبار", "الجمهورية العربية", "ܫܠܡܐ ܥܠܡܐ", "ދިވެހިރާއްޖެ", "mixed ascii + עברית + العربية together",
    "Привет, мир! Это проверка.", "Съешь ещё этих мягких французских булок",
]


def planted_image(n, seed, encodings=("utf-8", "utf-16-le", "utf-16-be"), every=65536):
    """BASELINE.md 'disk image': the background with a record planted every 64 KiB, 12..400 chars,
    in each encoding in turn; some straddle 128-byte windows, 4096-byte slices and 256 KiB
    sub-chunks, some are exactly 64 chars long."""
    rng = random.Random(seed)
    img = bytearray(sxo.background(0, n, SEED))
    k = 0
    for base in range(every, n - 2048, every):
        enc = encodings[k % len(encodings)]
        pool = [t for t in CORPUS if "и" in t or "е" in t] if enc == "koi8-r" else CORPUS
        text = rng.choice(pool)
        want_chars = rng.choice([12, 20, 63, 64, 64, 65, 100, 128, 129, 400, rng.randrange(12, 400)])
        while len(text) < want_chars:
            text += " " + rng.choice(pool)
        text = text[:want_chars]
        try:
            rec = text.encode(enc)
        except UnicodeEncodeError:
            rec = text.encode(enc, "ignore") if enc in ("big5hkscs", "euc_jp") else text.encode("utf-8")
        mode = k % 5
        if mode == 0:
            off = base - len(rec) // 2            # straddles a 64 KiB (hence 4096 / 128 / every 4th a 256 KiB) edge
        elif mode == 1:
            off = base + 4096 * rng.randrange(1, 8) - rng.randrange(1, 8)   # a slice edge
        elif mode == 2:
            off = base + 128 * rng.randrange(1, 200) - rng.randrange(1, 40)  # a window edge
        else:
            off = base + rng.randrange(0, every - 2048)
        if enc.startswith("utf-16"):
            off &= ~1
        img[off:off + len(rec)] = rec
        img[off - 1:off] = b"\x00" if not enc.startswith("utf-16") else img[off - 1:off]
        k += 1
    return bytes(img)


def scan_text(ms, data, **kw):
    sc = sx.Scanner(ms, device=0, **kw)
    d = sc.alloc(len(data)); sc.upload(d, data)
    res = sc.scan_device(d, len(data), file_id=1)
    out = sx.OUTPUT_BOM + res.printed(n_inputs=1, radix="x") + b"\n"
    n = len(res)
    res.free(); sc.free(d); sc.close()
    return out, n


def test_c1_ascii_1mib_full_text_diff():
    ms = product_missions(encodings=["ascii"], chars_min="4")
    host = sxo.background(0, 1 << 20, SEED)
    got, n = scan_text(ms, host)
    assert got == sxo.run_cli(ms, [host], radix="x")
    assert 9000 < n < 15000          # ~11.8 k findings per MiB


def test_c2_utf8_slice_text_diff_and_whole_size_invariants():
    ms = product_missions(encodings=["utf-8"], chars_min="10")
    pre = 64 << 20
    host = sxo.background(0, pre, SEED)
    got, n = scan_text(ms, host)
    assert got == sxo.run_cli(ms, [host], radix="x")
    assert 40 * 64 < n < 80 * 64      # ~59 findings per MiB
    # the full 4 GiB of the config, generated on the device: one call == two calls with carried state
    total = 4 << 30
    sc = sx.Scanner(ms, device=0)
    d = sc.alloc(total); sc.fill_background(d, 0, total, SEED)
    whole = sc.scan_device(d, total, file_id=1)
    n_whole = len(whole)
    whole_pos = [(s[0][i].position) for s in whole.segments() for i in range(0, s[1], max(1, s[1] // 1000))]
    whole.free(); sc.reset()
    half = total // 2
    a = sc.scan_device(d, half, file_id=1)
    b = sc.scan_device(ctypes.c_void_p(d.value + half), total - half, file_id=1)
    assert len(a) + len(b) == n_whole
    assert whole_pos == sorted(whole_pos)
    a.free(); b.free(); sc.free(d); sc.close()


def test_c3_run_records_equal_oracle_on_1gib_prefix():
    ms = product_missions(encodings=["utf-8", "utf-16le", "utf-16be"], chars_min="10", unicode_block_filter="African")
    n = 1 << 30
    host = sxo.background(0, n, SEED)
    sc = sx.Scanner(ms, device=0)
    d = sc.alloc(n); sc.fill_background(d, 0, n, SEED)
    for k, m in enumerate(ms):
        got = sc.device_runs(k, d, n, stream_parity=0, min_chars=10)
        want = sxo.runs(m, host, stream_parity=0, min_chars=10, cap=1 << 22)
        assert got == want, (k, len(got), len(want))
    sc.free(d); sc.close()


@pytest.mark.parametrize("device_replay", [None, False])
def test_c3ii_disk_image_256mib_full_text_diff(device_replay):
    ms = product_missions(encodings=["utf-8", "utf-16le", "utf-16be"], chars_min="10", unicode_block_filter="African")
    img = planted_image(256 << 20, 3)
    got, n = scan_text(ms, img, device_replay=device_replay)
    want = sxo.run_cli(ms, [img], radix="x")
    assert got == want
    assert n > 4000                   # every planted record that passes the filters is there


@pytest.mark.parametrize("device_replay", [None, False])
def test_c5_six_missions_disk_image_full_text_diff(device_replay):
    """BASELINE config 5: all six missions, per-encoding filters in the reference's own syntax (SURVEY 8a, C5
    note), on a planted image with Big5 (HKSCS), EUC-JP and KOI8-R records next to the UTF ones."""
    ms = product_missions(encodings=["utf-8,,,African", "utf-16le,,,African", "utf-16be,,,African", "big5,,,Cjk",
                                     "euc-jp,,,Asian", "koi8-r,,,Cyrillic"], chars_min="10")
    from test_dbcs import TEXT
    CORPUS.extend([TEXT["big5"][:40], TEXT["euc-jp"][:40], TEXT["big5"][20:45], TEXT["euc-jp"][30:70]])
    try:
        img = planted_image(32 << 20, 5, encodings=("utf-8", "utf-16-le", "utf-16-be", "koi8-r", "big5hkscs", "euc_jp"))
    finally:
        del CORPUS[-4:]
    got, n = scan_text(ms, img, device_replay=device_replay)
    assert got == sxo.run_cli(ms, [img], radix="x")
    assert b"(f KOI8-R)" in got and b"(d Big5)" in got and b"(e EUC-JP)" in got and n > 1000
