"""Parity at benchmark scale: the headline configuration (C3, 64 GiB, three missions) and C2 (4 GiB) are scanned once
on the device, then pseudo-random 64 MiB windows of the buffer — half of them above 2^32, one ending at the last
byte — are regenerated on the host, scanned by the ORACLE with counter_offset = window start, and every finding
whose slice lies inside the window (a margin at both ends lets the oracle's fresh state settle) must be identical:
position, precision, `+`, Mission, string, slice_index.  This is what would catch a 32-bit truncation of offsets,
slice indices, string offsets or record-slot arithmetic that a 1 GiB prefix cannot show."""
import ctypes
import random
import threading

import numpy as np
import pytest

import refconfig as rc
import stringsext_amd as sx
import sxo_binding as sxo
from test_gpu_baseline_configs import CORPUS, SEED, product_missions

pytestmark = pytest.mark.gpu
WINDOW = 64 << 20
MARGIN = 64 << 10
FDT = np.dtype({"names": ["position", "str_off", "str_len", "precision", "completes", "mission_id", "slice_index"],
                "formats": ["<u8", "<u4", "<u4", "u1", "u1", "u1", "<u4"], "offsets": [0, 8, 12, 16, 17, 18, 24], "itemsize": 32})


def product_findings_by_slice(res):
    """numpy views of the result's segments (no per-finding Python objects for millions of findings)"""
    out = []
    for fp, n, ap, alen in res.segment_pointers():
        if n:
            f = np.ctypeslib.as_array(ctypes.cast(fp, ctypes.POINTER(ctypes.c_uint8)), shape=(n * 32,)).view(FDT)
            a = np.ctypeslib.as_array(ap, shape=(alen,)) if alen else np.zeros(0, np.uint8)
            out.append((f, a))
    return out


def window_findings(segs, lo_slice, hi_slice):
    got = []
    for f, a in segs:
        i0, i1 = np.searchsorted(f["slice_index"], [lo_slice, hi_slice])   # findings are in slice order
        for r in f[i0:i1]:
            s = bytes(a[int(r["str_off"]):int(r["str_off"]) + int(r["str_len"])]).decode("utf-8")
            got.append((int(r["position"]), sx.PRECISION[int(r["precision"])], s, bool(r["completes"]), int(r["mission_id"]), int(r["slice_index"])))
    return got


def oracle_window(ms, host, ws):
    """the reference loop over one window that starts at stream offset ws (a multiple of 4096): one thread per mission"""
    per = [None] * len(ms)

    def run(k):
        sc = sxo.Scanner(dict(ms[k], counter_offset=ms[k]["counter_offset"] + ws))
        rows = []
        for si, off in enumerate(range(0, len(host), 4096)):
            for i, f in enumerate(sc.scan(host[off:off + 4096], file_id=1)):
                rows.append((ws // 4096 + si, f["position"], k, i, f))
        per[k] = rows
    th = [threading.Thread(target=run, args=(k,)) for k in range(len(ms))]
    [t.start() for t in th]
    [t.join() for t in th]
    rows = sorted((r for p in per for r in p), key=lambda t: t[:4])
    return [(f["position"], f["precision"], f["s"], f["completes"], k, si) for si, _, k, _, f in rows]


def patches_for(total, rng, n=600):
    """planted records (C3(ii)): at 2^32, 2^33, 2^35 and the buffer end, at slice / window / sub-chunk edges, some
    exactly 64 chars long"""
    spots = [1 << 32, (1 << 32) + 4096, 1 << 33, 1 << 35, total - 4096, total - 200, (1 << 32) - 262144, (3 << 32) + 131072]
    spots += [rng.randrange(1 << 20, total - (1 << 20)) // 128 * 128 for _ in range(n)]
    out = []
    for i, at in enumerate(spots):
        enc = ("utf-8", "utf-16-le", "utf-16-be")[i % 3]
        text = rng.choice(CORPUS)
        want = rng.choice([12, 20, 63, 64, 65, 100, 129, 300])
        while len(text) < want:
            text += " " + rng.choice(CORPUS)
        rec = b"\x00\x00" + text[:want].encode(enc) + b"\x00\x00"
        off = max(0, min(total - len(rec), at - rng.choice([0, 1, 7, len(rec) // 2, len(rec) - 3]))) & ~1
        out.append((off, rec))
    out.sort()
    keep, end = [], -1      # no overlaps: a patch is what the host regenerates
    for off, rec in out:
        if off >= end:
            keep.append((off, rec)); end = off + len(rec)
    return keep


def regenerate(ws, n, patches):
    host = bytearray(sxo.background(ws, n, SEED))
    for off, rec in patches:
        if off + len(rec) > ws and off < ws + n:
            a, b = max(off, ws), min(off + len(rec), ws + n)
            host[a - ws:b - ws] = rec[a - off:b - off]
    return bytes(host)


def check_windows(ms, total, planted, n_windows, seed, first_starts=(), scans=1):
    rng = random.Random(seed)
    sc = sx.Scanner(ms, device=0)
    d = sc.alloc(total)
    sc.fill_background(d, 0, total, SEED)
    patches = patches_for(total, rng) if planted else []
    for off, rec in patches:
        sc.upload(ctypes.c_void_p(d.value + off), rec)
    res = sc.scan_device(d, total, file_id=1)
    for _ in range(scans - 1):     # again, with what the first scan learnt (Mission order, density, output size)
        res.free(); sc.reset()
        res = sc.scan_device(d, total, file_id=1)
    try:
        segs = product_findings_by_slice(res)
        assert sum(len(f) for f, _ in segs) == len(res)
        starts = [total - WINDOW] + list(first_starts)               # ends at the last byte
        if total > (1 << 32):
            starts += [(1 << 32) - WINDOW // 2, (1 << 33) - 4096]     # across 2^32 and 2^33
        while len(starts) < n_windows:
            lo = (1 << 32) if (len(starts) % 2 == 0 and total > (1 << 33)) else 0   # at least half start above 2^32
            starts.append(rng.randrange(lo, total - WINDOW) // 4096 * 4096)
        if planted:   # windows around planted records above 2^32, too
            starts[3:7] = [max(0, min(total - WINDOW, off - WINDOW // 2)) // 4096 * 4096 for off, _ in patches[-5:-1]]
        compared = 0
        for ws in starts:
            host = regenerate(ws, WINDOW, patches)
            want = oracle_window(ms, host, ws)
            at_end = ws + WINDOW == total
            lo_slice, hi_slice = (ws + MARGIN) // 4096, (ws + WINDOW - (0 if at_end else MARGIN)) // 4096
            want = [t for t in want if lo_slice <= t[5] < hi_slice]
            got = window_findings(segs, lo_slice, hi_slice)
            assert got == want, (hex(ws), len(got), len(want), next(((a, b) for a, b in zip(got, want) if a != b), None))
            compared += len(want)
        return compared, len(res)
    finally:
        res.free(); sc.free(d); sc.close()


C3 = dict(encodings=["utf-8", "utf-16le", "utf-16be"], chars_min="10", unicode_block_filter="African")


@pytest.mark.parametrize("planted", [False, True], ids=["c3i-background", "c3ii-planted"])
def test_c3_64gib_windows_equal_oracle(planted):
    ms = product_missions(**C3)
    compared, total = check_windows(ms, 64 << 30, planted, 16, 64 + planted)
    assert total > 2_000_000 and compared > 16 * 1500      # ~35 findings per MiB on the background


def test_c2_4gib_windows_equal_oracle():
    ms = product_missions(encodings=["utf-8"], chars_min="10")
    compared, total = check_windows(ms, 4 << 30, True, 6, 4)
    assert total > 200_000 and compared > 6 * 3000        # ~59 findings per MiB


C5 = dict(encodings=["utf-8,,,African", "utf-16le,,,African", "utf-16be,,,African", "big5,,,Cjk", "euc-jp,,,Asian", "koi8-r,,,Cyrillic"],
          chars_min="10")


def test_c5_64gib_windows_equal_oracle():
    """BASELINE config 5 at its full size: on the synthetic background the double-byte and KOI8-R Missions produce a flood
    (411 M findings, 19 GB) — the scale at which the outputs stay on the device, pass 1's cache takes tens of GiB and the
    merger runs on the device in several parts (one result segment each, str_off restarting per segment).  Windows across
    2^32 and 2^33, at the buffer's end and at a random place must equal the oracle finding by finding."""
    ms = product_missions(**C5)
    compared, total = check_windows(ms, 64 << 30, False, 4, 5)
    assert total > 300_000_000 and compared > 4 * 300_000


def test_c5_16gib_in_sequential_pieces_equals_oracle(monkeypatch):
    """The same flood in pieces of 5 GiB scanned one after the other (scan_common's sequential pieces: what a buffer with
    gigabytes of output is cut into, each piece's merged findings copied while the next piece is scanned and replayed; the
    double-byte Missions enter a piece with the decoder the piece in front left).  Windows across the piece boundaries."""
    monkeypatch.setenv("SX_SEQ_PIECE_MIB", "5120")
    ms = product_missions(**C5)
    cuts = [(5 << 30) - WINDOW // 2, (10 << 30) - WINDOW // 4, (15 << 30) - 3 * WINDOW // 4]
    compared, total = check_windows(ms, 16 << 30, False, 5, 55, first_starts=cuts, scans=2)
    assert total > 75_000_000 and compared > 5 * 300_000


def test_wave_path_6gib_windows_equal_oracle():
    """One string-dense Mission through the wave-cooperative stage B in slabs (csrc/sx_wave.cpp): 6 GiB, so that windows,
    slices and string offsets pass 2^31 and 2^32; the oracle's windows include both."""
    ms = product_missions(encodings=["koi8-r,,,Cyrillic"], chars_min="10")
    total = 6 << 30
    sc = sx.Scanner(ms, device=0)
    d = sc.alloc(total)
    sc.fill_background(d, 0, total, SEED)
    res = sc.scan_device(d, total, file_id=1)
    try:
        assert sc.stats().wave_windows == total // 128 - 1   # all but the buffer's first window (the host's)
        segs = product_findings_by_slice(res)
        assert len(segs) > 1 and sum(len(f) for f, _ in segs) == len(res) > 20_000_000
        small = 16 << 20
        for ws in ((1 << 31) - small // 2, (1 << 32) - small // 2, total - small, 5 * (1 << 30) + 4096 * 77):
            host = sxo.background(ws, small, SEED)
            want = oracle_window(ms, host, ws)
            at_end = ws + small == total
            lo_slice, hi_slice = (ws + MARGIN) // 4096, (ws + small - (0 if at_end else MARGIN)) // 4096
            want = [t for t in want if lo_slice <= t[5] < hi_slice]
            got = window_findings(segs, lo_slice, hi_slice)
            assert got == want, (hex(ws), len(got), len(want), next(((a, b) for a, b in zip(got, want) if a != b), None))
    finally:
        res.free(); sc.free(d); sc.close()


def test_wave_path_utf16_6gib_windows_equal_oracle(monkeypatch):
    """UTF-16LE through the wave kernels (round 4) at 6 GiB: unit grid, window numbers and string offsets beyond 2^31 / 2^32; random units
    hold lone surrogates every 32 units on average — the slow mode behind a pending high surrogate at every other window start"""
    monkeypatch.setenv("SX_WAVE_REPLAY", "1")
    ms = product_missions(encodings=["utf-16le,,,Cjk"], chars_min="4")
    total = 6 << 30
    sc = sx.Scanner(ms, device=0)
    d = sc.alloc(total)
    sc.fill_background(d, 0, total, SEED)
    res = sc.scan_device(d, total, file_id=1)
    try:
        assert sc.stats().wave_windows == total // 128 - 1   # all but the buffer's first window (the host's): no wavefront gave up
        segs = product_findings_by_slice(res)
        assert sum(len(f) for f, _ in segs) == len(res) > 5_000_000
        small = 16 << 20
        for ws in ((1 << 31) - small // 2, (1 << 32) - small // 2, total - small, 5 * (1 << 30) + 4096 * 77):
            host = sxo.background(ws, small, SEED)
            want = oracle_window(ms, host, ws)
            at_end = ws + small == total
            lo_slice, hi_slice = (ws + MARGIN) // 4096, (ws + small - (0 if at_end else MARGIN)) // 4096
            want = [t for t in want if lo_slice <= t[5] < hi_slice]
            got = window_findings(segs, lo_slice, hi_slice)
            assert got == want, (hex(ws), len(got), len(want), next(((a, b) for a, b in zip(got, want) if a != b), None))
    finally:
        res.free(); sc.free(d); sc.close()
