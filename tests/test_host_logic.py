"""CPU tests of the product's host side: the C-ABI library loads and exports every symbol of
include/stringsext_amd.h, refuses to run without a GPU, and its replay stage (fed with the
run records the ORACLE's sequential decoder computes) reproduces the reference's golden
outputs, unit-test known answers and the oracle's full scan on adversarial inputs."""
import ctypes
import os
import random
import re

import pytest

import refconfig as rc
import stringsext_amd as sx
import sxo_binding as sxo
from golden import unit_kats as K
from product_harness import ProductScanner, run_cli_product
from test_oracle_golden import CLI_CASES, rd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "stringsext_amd.h")).read()
    declared = set(re.findall(r"\b(sx_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"sx_scan_device"} - set(sx.EXPORTS)  # nothing to subtract; keeps intent explicit
    L = sx.lib()
    for name in declared:
        assert hasattr(L, name), name
    assert set(sx.EXPORTS) == declared
    assert L.sx_abi_version() == 4


def test_no_gpu_means_no_scan():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(sx.SxError) as e:
        sx.Scanner([rc.mission()], device=0)
    assert e.value.code == sx.SX_E_NO_DEVICE
    sc = sx.Scanner([rc.mission()], device=sx.SX_HOST_ONLY)
    with pytest.raises(sx.SxError) as e:
        sc.scan(b"no device, no scan")
    assert e.value.code == sx.SX_E_STATE


@pytest.mark.parametrize("expected,flags,inputs", CLI_CASES, ids=[c[0] for c in CLI_CASES])
@pytest.mark.parametrize("chunk", [None, 4096, 8192])
def test_replay_reproduces_cli_goldens(expected, flags, inputs, chunk):
    out = run_cli_product(rc.missions(**flags), [rd(i) for i in inputs], radix="x", chunk_bytes=chunk)
    assert out == rd(expected)


@pytest.mark.parametrize("kat", K.SCAN_KATS, ids=[k["name"] for k in K.SCAN_KATS])
def test_replay_scan_known_answers(kat):
    sc = ProductScanner(kat["mission"])
    for i, call in enumerate(kat["calls"]):
        got = sc.scan(call["input"], file_id=0, is_last=call["is_last"])
        slim = [dict(position=f["position"], precision=f["precision"], s=f["s"]) for f in got]
        if "findings" in call:
            assert slim == call["findings"], (kat["name"], i)
        if "findings_prefix" in call:
            assert slim[:len(call["findings_prefix"])] == call["findings_prefix"], (kat["name"], i)
        if "n_findings_not" in call:
            assert len(slim) != call["n_findings_not"]


def test_replay_merger_known_answer():
    k = K.MERGER_KAT
    ms = rc.missions(**k["flags"])
    sc = sx.Scanner(ms, device=sx.SX_HOST_ONLY)
    from product_harness import oracle_runs_for_chunk
    res = sc.replay_runs(k["input"], oracle_runs_for_chunk(ms, k["input"], 0), file_id=0, is_last=True)
    got = [(f["s"], f["position"], f["precision"], f["mission_id"]) for f in res.findings()]
    assert got == k["merged"]


# ------------------------------------------------------------------------------------------
# differential: sparse replay == the oracle's full scan
# ------------------------------------------------------------------------------------------
WORDS_ASCII = [b"/usr/lib/x86_64-linux-gnu/libfoo.so.1", b"C:\\Windows\\System32\\drivers\\etc\\hosts", b"hello world",
               b"The quick brown fox jumps over the lazy dog. " * 5, b"A" * 64, b"B" * 63, b"C" * 65, b"D" * 128,
               b"x" * 300, b"key=value; path=/;", b"abc", b"abcd", b"0123456789"]
WORDS_UNI = ["Բարեւ աշխարհ ողջույն", "שלום עולם מה שלומך היום", "مرحبا بالعالم كيف حالك", "Привет, мир! Как дела?",
             "Καλημέρα κόσμε", "žluťoučký kůň úpěl ďábelské ódy", "日本語のテキスト文字列", "𝔘𝔫𝔦𝔠𝔬𝔡𝔢 𝔱𝔢𝔵𝔱 😀😀😀",
             "ÄÖÜäöüß" * 12, "ՀայերենՀայերենՀայերեն" * 4]


def synth(rng, size, density, encodings=("utf-8", "utf-16le", "utf-16be", "latin")):
    """Random bytes with planted strings in several encodings, some at slice/window edges."""
    buf = bytearray(rng.randbytes(size))
    pos = 0
    while pos < size:
        pos += int(rng.expovariate(density)) + 1
        if rng.random() < 0.25:  # snap near a 4096 / 128 boundary
            g = rng.choice([4096, 128, 1024])
            pos = (pos // g + 1) * g - rng.randrange(0, 24)
        w = rng.choice(WORDS_ASCII) if rng.random() < 0.5 else rng.choice(WORDS_UNI).encode("utf-8")
        enc = rng.choice(encodings)
        if enc == "utf-16le":
            w = w.decode("utf-8").encode("utf-16-le")
        elif enc == "utf-16be":
            w = w.decode("utf-8").encode("utf-16-be")
        elif enc == "latin":
            w = w.decode("utf-8").encode("koi8-r", "replace")
        if pos < 0 or pos + len(w) > size:
            continue
        buf[pos:pos + len(w)] = w
        pos += len(w)
    return bytes(buf)


def soup(rng, size):
    """Adversarial byte soup: leads, boundary continuations, surrogate halves, short ASCII."""
    alphabet = [0x41, 0x42, 0x20, 0x7E, 0x7F, 0x00, 0x0A, 0x80, 0x8F, 0x90, 0x9F, 0xA0, 0xBF, 0xC0, 0xC1, 0xC2,
                0xD5, 0xDF, 0xE0, 0xE1, 0xED, 0xEE, 0xF0, 0xF1, 0xF4, 0xF5, 0xFF, 0xD8, 0xDB, 0xDC, 0xDD, 0x05, 0x06]
    return bytes(rng.choice(alphabet) for _ in range(size))


OPTION_SETS = [
    dict(chars_min="4"),
    dict(chars_min="10", unicode_block_filter="African"),
    dict(chars_min="5", output_line_len="10"),
    dict(chars_min="3", output_line_len="6", grep_char="47"),
    dict(chars_min="12", output_line_len="8"),
    dict(chars_min="4", same_unicode_block=True, unicode_block_filter="All"),
    dict(chars_min="6", output_line_len="30", ascii_filter="All-Ctrl+Wsp", unicode_block_filter="All"),
    dict(chars_min="2", ascii_filter="0x7ffffffe000000007ffffffe00000000", unicode_block_filter="Latin"),
    dict(chars_min="1", output_line_len="7", unicode_block_filter="Cyrillic"),
    dict(chars_min="8", unicode_block_filter="Uncommon", ascii_filter="None"),
    dict(chars_min="0", output_line_len="9"),   # -n 0: two rejected chars in a row end a decoder call's iteration (helper.rs:317,343)
    dict(chars_min="0", unicode_block_filter="All", grep_char="0x65"),
]
ENC_SETS = [["utf-8"], ["ascii"], ["utf-16le"], ["utf-16be"], ["koi8-r"], ["utf-8", "utf-16le", "utf-16be"],
            ["ascii", "utf-8", "koi8-r"], ["utf-8,3,All,All", "utf-16le,,,Asian", "ascii,5"]]


def _cases():
    rng = random.Random(20240928)
    cases = []
    for i, opts in enumerate(OPTION_SETS):
        for j, encs in enumerate(ENC_SETS):
            if (i + j) % 3 == 0:  # a third of the grid keeps the CPU suite short
                cases.append((opts, encs, rng.randrange(1 << 30)))
    return cases


@pytest.mark.parametrize("opts,encs,seed", _cases(), ids=lambda v: str(v)[:40])
def test_sparse_replay_equals_full_scan(opts, encs, seed):
    rng = random.Random(seed)
    ms = rc.missions(encodings=encs, **opts)
    kind = rng.choice(["synth_dense", "synth_sparse", "soup", "odd_files", "text"])
    if kind == "synth_dense":
        files = [synth(rng, 40000, 1 / 200)]
    elif kind == "synth_sparse":
        files = [synth(rng, 120000, 1 / 3000)]
    elif kind == "soup":
        files = [soup(rng, 30000)]
    elif kind == "odd_files":  # odd lengths shift the UTF-16 unit parity and restart the slice grid
        files = [synth(rng, 4096 * 2 + 1, 1 / 150), synth(rng, 8193 + 4096, 1 / 150), soup(rng, 777), b"",
                 synth(rng, 20001, 1 / 300)]
    else:
        files = [rd("input1"), rd("input2")[:30001]]
    want = sxo.run_cli(ms, files, radix="x")
    for chunk in (None, 4096, 12288):
        got = run_cli_product(ms, files, radix="x", chunk_bytes=chunk)
        assert got == want, (kind, chunk)


def test_flush_at_eof_path():
    ms = rc.missions(encodings=["utf-8", "utf-16le"], chars_min="4", output_line_len="8")
    rng = random.Random(5)
    for _ in range(10):
        f = synth(rng, rng.randrange(1, 9000), 1 / 100)
        assert run_cli_product(ms, [f], radix="d", flush_at_eof=True) == sxo.run_cli(ms, [f], radix="d", flush_at_eof=True)


def test_print_variants():
    ms = rc.missions(encodings=["ascii", "utf-8"], chars_min="5")
    f = [rd("input1")]
    for radix in (None, "x", "d", "o"):
        for nometa in (False, True):
            assert run_cli_product(ms, f, radix=radix, no_metadata=nometa) == sxo.run_cli(ms, f, radix=radix, no_metadata=nometa)


def test_parallel_parts_and_stitch():
    """Inputs large enough for several replay parts (4 MiB each), with strings planted across
    the part boundaries so that the speculative parts must be verified and repaired."""
    rng = random.Random(99)
    n = 17 << 20
    base = bytearray(sxo.background(0, n))
    part = (n // 4 + 4095) // 4096 * 4096  # replay_plan: 4 parts for 17 MiB
    long_ascii = bytes(rng.choice(b"abcdefghijklmnopqrstuvwxyz /._-") for _ in range(9000))
    heb = ("שלום עולם " * 700).encode("utf-8")
    heb16 = ("שלום עולם " * 300).encode("utf-16-le")
    for k in range(1, 8):
        b = k * (n // 8) // 4096 * 4096
        for off, blob in ((-4000, long_ascii), (2 << 20, heb), (-100, heb16), (3 << 20, long_ascii[:63]),
                          ((1 << 20) - 64, long_ascii[:64]), ((1 << 20) + 128 - 10, long_ascii[:20])):
            p = b + off + rng.randrange(0, 3)
            if 0 <= p and p + len(blob) < n:
                base[p:p + len(blob)] = blob
    for k in (1, 2, 3):  # exactly at the real part boundaries
        p = k * part - 5000
        base[p:p + len(long_ascii)] = long_ascii
    data = bytes(base)
    for flags in (dict(encodings=["utf-8", "utf-16le", "utf-16be"], chars_min="10", unicode_block_filter="African"),
                  dict(encodings=["ascii"], chars_min="12", output_line_len="20"),
                  dict(encodings=["utf-8"], chars_min="70", output_line_len="64"),
                  dict(encodings=["utf-8"], chars_min="6", output_line_len="32", grep_char="47")):
        ms = rc.missions(**flags)
        assert run_cli_product(ms, [data], radix="x") == sxo.run_cli(ms, [data], radix="x"), flags


def test_fuzz_sparse_replay():
    """Many small adversarial inputs x random option combinations: the sparse replay must
    equal the oracle's full scan byte for byte."""
    rng = random.Random(424242)
    enc_pool = ["utf-8", "ascii", "utf-16le", "utf-16be", "koi8-r", "windows-1252"]
    ubf_pool = [None, "African", "All", "Common", "Latin", "Cyrillic", "None", "Arabic", "0x0010400000000000"]
    af_pool = [None, "All", "All-Ctrl+Wsp", "None", "Wsp"]
    alphabet_hi = list(range(0x20, 0x7F)) * 3 + [0xC3, 0xA9, 0xD7, 0x90, 0xD5, 0xB1, 0xE2, 0x82, 0xAC, 0xF0, 0x9F,
                                                 0x98, 0x80, 0x00, 0x0A, 0xFF, 0xC0, 0xED, 0xA0, 0xD8, 0x00, 0xDC]
    for it in range(160):
        n = rng.choice([1, 2, 3, 4, 5, 7, 10, 16, 33, 64, 70])
        q = rng.choice([6, 7, 8, 10, 16, 32, 64])
        encs = rng.sample(enc_pool, rng.choice([1, 1, 2, 3]))
        flags = dict(encodings=encs, chars_min=str(n), output_line_len=str(q),
                     unicode_block_filter=rng.choice(ubf_pool), ascii_filter=rng.choice(af_pool),
                     same_unicode_block=rng.random() < 0.2,
                     grep_char=rng.choice([None, None, None, "47", "32", "101"]))
        ms = rc.missions(**flags)
        files = []
        for _ in range(rng.choice([1, 1, 2, 3])):
            size = rng.choice([0, 1, 5, 127, 128, 129, 4095, 4096, 4097, 9000, 20000])
            kind = rng.random()
            if kind < 0.3:
                d = bytes(rng.choice(alphabet_hi) for _ in range(size))
            elif kind < 0.6:
                d = synth(rng, size, 1 / 60) if size else b""
            elif kind < 0.8:
                d = soup(rng, size)
            else:  # long text-like stretches, cut by rare breaks
                d = bytes((rng.choice(b"abcdefghij klmnop/") if rng.random() > 0.01 else rng.choice([0, 0xFF, 0xC3]))
                          for _ in range(size))
            files.append(d)
        want = sxo.run_cli(ms, files, radix="x")
        got = run_cli_product(ms, files, radix="x", chunk_bytes=rng.choice([None, 4096, 8192]))
        assert got == want, (it, flags, [len(f) for f in files])


def test_carried_state_across_many_small_files():
    """Text-heavy streams cut into many small files (every cut restarts the slice grid and
    hands a leftover / cut-string flag / partial character to the next file)."""
    rng = random.Random(31337)
    words = [w.decode("utf-8", "ignore") for w in WORDS_ASCII] + WORDS_UNI
    for it in range(40):
        enc = rng.choice(["utf-8", "utf-16le", "utf-16be"])
        text = "".join(rng.choice(words) + rng.choice(["\x00", "\x01\x02", "", " ", "\n"]) for _ in range(60))
        blob = text.encode({"utf-8": "utf-8", "utf-16le": "utf-16-le", "utf-16be": "utf-16-be"}[enc])
        blob = bytearray(blob)
        for _ in range(len(blob) // 200):  # sprinkle breaks
            blob[rng.randrange(len(blob))] = rng.choice([0xFF, 0xC0, 0x00, 0xD8, 0xDC])
        blob = bytes(blob)
        cuts = sorted(rng.sample(range(1, len(blob)), min(len(blob) - 1, rng.choice([3, 10, 25]))))
        files = [blob[a:b] for a, b in zip([0] + cuts, cuts + [len(blob)])]
        ms = rc.missions(encodings=[enc, "ascii"], chars_min=str(rng.choice([3, 5, 9, 14])),
                         output_line_len=str(rng.choice([6, 9, 16, 64])),
                         unicode_block_filter=rng.choice(["All", "Common", "African"]),
                         same_unicode_block=rng.random() < 0.2)
        assert run_cli_product(ms, files, radix="x") == sxo.run_cli(ms, files, radix="x"), (it, enc, len(files))


def test_character_straddling_a_chunk_boundary_starts_a_minimal_string():
    """A run of exactly chars_min chars whose first character straddles the chunk boundary: neither
    chunk's run finder sees a long run (the first chunk cannot count the incomplete character, the
    second sees chars_min - 1), so only the exact carried state — here: the decoder's pending byte —
    can make the next chunk's first windows be replayed (found by tools/gpu_fuzz.py)."""
    ms = rc.missions(encodings=["utf-8"], output_line_len="8", ascii_filter="Wsp", unicode_block_filter="0x001ffffffffffffc")
    for lead_in in (1, 2, 3):
        for word in ("Ждя ", "€дя ", "😀дя "):
            w = word.encode("utf-8")
            first = len(word[0].encode("utf-8"))
            if lead_in >= first:
                continue
            data = b"\xff" * (4096 * 3 - lead_in) + w + b"\x00" + b"0Zc" * 40 + b"\xff" * 5000
            want = sxo.run_cli(ms, [data], radix="x")
            assert word.strip().encode("utf-8") in want
            for chunk in (4096, 8192, None):
                assert run_cli_product(ms, [data], radix="x", chunk_bytes=chunk) == want, (lead_in, word, chunk)


def test_same_unicode_block_remembers_the_leftovers_last_multibyte_lead():
    """-r: SplitStr re-scans the leftover and keeps the lead byte of its last multi-byte char; that decides
    where the next stretch of the same decoder call is cut (src/helper.rs:279-292).  A replay region that
    begins right after such a leftover must re-derive it, not just "some accepted char" (found by
    tools/gpu_fuzz.py: a 2-char leftover 'Ѕ/' in front of '+ASy' + U+009B in ISO-8859-5)."""
    ms = rc.missions(encodings=["iso-8859-5"], chars_min="4", output_line_len="6", ascii_filter="All-Ctrl",
                     unicode_block_filter="Common", same_unicode_block=True)
    W = 12
    for pad in range(0, 3):
        body = b"\xa5/" + b"\x1a+ASy\x9b\r\xaa\x91X\x88\xcf" + b"\x00" * 40
        for k in (5, 40, 339):
            data = b"\x00" * (W * k - 2 - pad) + b"/" * pad + body + b"\x01" * 3000
            want = sxo.run_cli(ms, [data], radix="x")
            assert b"+ASy" in want
            assert run_cli_product(ms, [data], radix="x") == want, (pad, k)
            assert run_cli_product(ms, [data], radix="x", chunk_bytes=4096) == want, (pad, k)


def test_same_unicode_block_region_entry_without_exact_state_keeps_the_unit_grid():
    """-r looks further back when it re-derives a region's entry state; the replay parts behind the first
    one (and the regions behind the device's) have no exact state to start from, and must not take the
    "no state known" marker for one: UTF-16 was decoded one byte off the unit grid (found by
    tools/gpu_fuzz.py: utf-16be -n 4 -q 30 -r -u Asian, strings of U+4200 that are not there)."""
    n = 17 << 20
    part = (n // 4 + 4095) // 4096 * 4096  # replay_plan: 4 parts for 17 MiB
    base = bytearray(sxo.background(0, n))
    text = "一丁七万丈三上下不与" * 4  # one lead byte (E4): one "unicode block" for -r
    for k in (1, 2, 3):
        for delta, enc in ((10, "utf-16-be"), (4096 + 20, "utf-16-le")):
            blob = text.encode(enc)
            base[k * part + delta:k * part + delta + len(blob)] = blob
    # enough long runs elsewhere for the replay to be split at all (>= 512 per part)
    rng = random.Random(5)
    for _ in range(4000):
        p = rng.randrange(0, n - 200) // 2 * 2
        if all(abs(p - k * part) > 20000 for k in (1, 2, 3)):
            base[p:p + 40] = "丐丑丒专且丕".encode("utf-16-be") * 3 + b"\xff\xff\xff\xff"
    data = bytes(base)
    for flags in (dict(encodings=["utf-16be"], chars_min="4", output_line_len="30", unicode_block_filter="Asian",
                       same_unicode_block=True),
                  dict(encodings=["utf-16le"], chars_min="4", unicode_block_filter="Asian", same_unicode_block=True)):
        ms = rc.missions(**flags)
        want = sxo.run_cli(ms, [data], radix="x")
        assert text[:8].encode("utf-8") in want
        assert run_cli_product(ms, [data], radix="x") == want, flags


NEW_SINGLE_BYTE = ["iso-8859-3", "iso-8859-4", "iso-8859-6", "iso-8859-7", "iso-8859-8", "iso-8859-8-i", "iso-8859-10", "iso-8859-13",
                   "iso-8859-14", "iso-8859-16", "koi8-u", "macintosh", "windows-874", "windows-1250", "windows-1253", "windows-1254",
                   "windows-1255", "windows-1256", "windows-1257", "windows-1258", "x-mac-cyrillic"]


@pytest.mark.parametrize("enc", NEW_SINGLE_BYTE)
def test_remaining_single_byte_encodings_replay_equals_full_scan(enc):
    """the other WHATWG single-byte decoders go through the same table-driven path; text in the
    encoding itself (every defined high byte), undefined bytes as breaks, and random bytes."""
    t, n = sx.decoder_table(rc.ENC_IDS[enc])
    tab = [t[i] for i in range(n)]
    defined = [0x80 + i for i, v in enumerate(tab) if v and v >= 0xA0]
    undefined = [0x80 + i for i, v in enumerate(tab) if not v] or [0x00]
    rng = random.Random(zlib_seed(enc))
    words = [bytes(rng.choice(defined) for _ in range(rng.randrange(1, 30))) for _ in range(60)] + [b"plain ascii", b" ", b"x"]
    text = bytearray()
    while len(text) < 60_000:
        text += rng.choice(words)
        r = rng.random()
        if r < 0.3: text += bytes([rng.choice(undefined)])
        elif r < 0.5: text += b"\x00" * rng.randrange(1, 200)
        elif r < 0.6: text += rng.randbytes(rng.randrange(1, 64))
    data = bytes(text) + rng.randbytes(20_000)
    for flags in (dict(encodings=[enc], chars_min="4", unicode_block_filter="All"),
                  dict(encodings=[enc, "utf-8"], chars_min="3", output_line_len="16", unicode_block_filter="All-Asian", same_unicode_block=True)):
        ms = rc.missions(**flags)
        want = sxo.run_cli(ms, [data], radix="x")
        assert len(want) > 2000
        assert run_cli_product(ms, [data], radix="x") == want, flags
        assert run_cli_product(ms, [data], radix="x", chunk_bytes=8192) == want, flags


def zlib_seed(s):
    import zlib
    return zlib.crc32(s.encode())


def test_ctypes_structs_have_the_headers_layout(tmp_path):
    """The Python shim restates the C-ABI's structs by hand: a C program that includes include/stringsext_amd.h prints every
    struct's size and the offset of its last field; the ctypes classes must agree (sx_stats only ever grows at its end)."""
    import ctypes as C
    import os
    import subprocess
    import stringsext_amd as sx
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    pairs = [("sx_mission", sx.Mission), ("sx_finding", sx.Finding), ("sx_run", sx.Run), ("sx_stats", sx.Stats), ("sx_options", sx.Options),
             ("sx_cli_flags", sx.CliFlags), ("sx_enc_opt", sx.EncOpt), ("sx_finding16", sx.Finding16), ("sx_segment_info", sx.SegmentInfo)]
    src = tmp_path / "layout.c"
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "stringsext_amd.h"', 'int main(void) {']
    for cname, cls in pairs:
        last = cls._fields_[-1][0]
        lines.append(f'    printf("{cname} %zu %zu\\n", sizeof({cname}), offsetof({cname}, {last}));')
    lines += ['    return 0;', '}']
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c11", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)])
    out = subprocess.check_output([str(exe)], text=True).split("\n")
    for (cname, cls), line in zip(pairs, out):
        name, size, off = line.split()
        assert name == cname
        assert (int(size), int(off)) == (C.sizeof(cls), getattr(cls, cls._fields_[-1][0]).offset), cname
