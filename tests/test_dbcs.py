"""Big5 and EUC-JP (BASELINE config 5; reference: help.rs:56-57 lists them, mission.rs:681 resolves them through
Encoding::for_label, the decoders are encoding_rs').  No reference test decodes a legacy multi-byte encoding
(parity unpinned), so the oracle's decoders are pinned here on two second sources —
CPython's codecs and hand-derived vectors of the WHATWG "Big5 decoder" / "EUC-JP decoder" algorithms — and the
product's host side (same code as the device's: sx_codec_core.hpp, sx_replay_core.hpp) is compared with the
oracle on text, binary soups and chunk boundaries."""
import random
import zlib

import pytest

import refconfig as rc
import stringsext_amd as sx
import sxo_binding as sxo
from product_harness import run_cli_product

ZH = ("這是一個測試字串，用來檢查大五碼解碼器。香港增補字符集：𠄌𠝹 Ê̄ê̌ 結束。"
      "天地玄黃宇宙洪荒日月盈昃辰宿列張寒來暑往秋收冬藏閏餘成歲律呂調陽")
JA = ("これは日本語のテスト文字列です。ﾊﾝｶｸｶﾅ ① 丂丄 漢字かな交じり文。"
      "いろはにほへとちりぬるをわかよたれそつねならむうゐのおくやまけふこえてあさきゆめみしゑひもせす")
KO = ("한국어 텍스트 테스트 문자열입니다. 똠방각하 펲시콜라 힣 ABC 가나다라마바사아자차카타파하 "
      "동해물과 백두산이 마르고 닳도록 하느님이 보우하사 우리나라 만세")
ZHS = ("这是一个测试字符串，用来检查国标码解码器。四字节：𠀀𠀁𪚥 ǹ ḿ ₠ ㊣ 〇 € 1234567890 结束。81308130 9 "
       "天地玄黄宇宙洪荒日月盈昃辰宿列张寒来暑往秋收冬藏闰余成岁律吕调阳")
CODEC = {"big5": "big5hkscs", "euc-jp": "euc_jp", "shift_jis": "cp932", "euc-kr": "cp949", "gbk": "gb18030", "gb18030": "gb18030"}
TEXT = {"big5": ZH, "euc-jp": JA, "shift_jis": JA + " 髙﨑 ", "euc-kr": KO, "gbk": ZHS, "gb18030": ZHS + ZH}
ENCS = ["big5", "euc-jp", "shift_jis", "euc-kr", "gbk", "gb18030"]
ALL = "0xffffffffffffffff"


def one(enc, data, **kw):
    """all strings the oracle finds, as (position, precision, completes, text)"""
    kw.setdefault("chars_min", "1")
    ms = rc.missions(encodings=[enc], unicode_block_filter=ALL, ascii_filter="0xffffffffffffffffffffffffffffffff", **kw)
    sc = sxo.Scanner(ms[0])
    return [(f["position"], f["precision"], f["completes"], f["s"]) for f in sc.scan(data, is_last=True)]


# ---- the oracle's decoders against second sources -------------------------------------------------------------------

@pytest.mark.parametrize("enc", ENCS)
def test_oracle_decodes_valid_text_like_cpython(enc):
    """Every mapped two-byte (and, EUC-JP, three-byte) sequence on its own, framed by NULs: the oracle prints what
    CPython's codec decodes, wherever the two sources of the tables agree (tests/test_tables.py)."""
    codec = CODEC[enc]
    checked = 0
    seqs = []
    for a in range(0x81, 0xFF):
        for b in list(range(0x40, 0x7F)) + list(range(0x80 if enc in ("shift_jis", "euc-kr", "gbk", "gb18030") else 0xA1, 0xFF)):
            seqs.append(bytes([a, b]))
    if enc == "euc-jp":
        seqs += [bytes([0x8F, a, b]) for a in range(0xA1, 0xFF, 3) for b in range(0xA1, 0xFF)]
    if enc in ("gbk", "gb18030"):   # four-byte sequences: a sample of the BMP ranges and of the astral planes
        seqs += [bytes([0x81 + p // 12600, 0x30 + p // 1260 % 10, 0x81 + p // 10 % 126, 0x30 + p % 10])
                 for p in list(range(0, 39420, 7)) + list(range(189000, 1237576, 997))]
    for i in range(0, len(seqs), 400):
        blob = b"".join(s + b"\n" for s in seqs[i:i + 400])
        got = sxo.run_cli(rc.missions(encodings=[enc], chars_min="1", unicode_block_filter=ALL), [blob], no_metadata=True)
        lines = got[3:].decode("utf-8").split("\n")
        found = {l for l in lines if l}
        for s in seqs[i:i + 400]:
            try:
                want = s.decode(codec)
            except UnicodeDecodeError:
                continue
            if any(ord(c) < 0x80 for c in want):
                continue  # cp/hkscs decodes lead + ASCII as two chars where WHATWG has an error + ASCII
            if want in found:
                checked += 1
    assert checked > {"big5": 13000, "euc-jp": 8000, "shift_jis": 7000, "euc-kr": 16000, "gbk": 29000, "gb18030": 29000}[enc]  # the bulk of the index (patched cells differ by design)


def test_oracle_big5_follows_the_whatwg_algorithm_on_hand_derived_vectors():
    # (bytes, expected strings with the position of the decoder call that printed them)
    A = b"AB"
    # valid pair, trail in 40..7E and in A1..FE
    assert one("big5", b"\xa4\x40\xa4\xa1")[0][3] == "一丑"
    # lead + ASCII trail that does not map (0x81 0x41: below the index): error, the ASCII byte is read again
    assert [x[3] for x in one("big5", b"\x81AB")] == ["AB"]
    assert one("big5", b"\x81AB")[0][0] == 1          # the call after the error starts AT the 'A'
    # lead + non-ASCII invalid trail (0x80): both consumed, next call starts behind them
    assert one("big5", b"\xa4\x80AB")[0][0] == 2
    # lead + 0xFF: both consumed
    assert one("big5", b"\xa4\xffAB")[0][0] == 2
    # 0x80 and 0xFF alone are one-byte errors
    assert one("big5", b"\x80" + A)[0][0] == 1 and one("big5", b"\xff" + A)[0][0] == 1
    # after a lead ANY byte returns to neutral: a4 a4 40 = (a4 a4)(40) not (a4)(a4 40)
    assert [x[3] for x in one("big5", b"\xa4\xa4\x40")] == [b"\xa4\xa4".decode("big5") + "@"]
    # the four pointers with two code points
    assert one("big5", b"\x88\x62\x88\x64\x88\xa3\x88\xa5")[0][3] == "Ê̄Ê̌ê̄ê̌"
    # a lead at the very end with is_last: error, nothing printed after it
    assert [x[3] for x in one("big5", b"AB\xa4")] == ["AB"]
    # astral (HKSCS): one four-byte UTF-8 char
    assert one("big5", "𠄌".encode("big5hkscs"))[0][3] == "𠄌"


def test_oracle_euc_jp_follows_the_whatwg_algorithm_on_hand_derived_vectors():
    assert one("euc-jp", b"\xa4\xa2\xa4\xa4")[0][3] == "あい"
    assert one("euc-jp", b"\x8e\xb1\x8e\xdf")[0][3] == "ｱﾟ"            # 8E + A1..DF: half-width katakana
    assert one("euc-jp", b"\x8e\xe0AB")[0][0] == 2                      # 8E + E0: error, both consumed
    assert one("euc-jp", b"\x8eAB")[0] [0] == 1 and one("euc-jp", b"\x8eAB")[0][3] == "AB"  # ASCII trail read again
    assert one("euc-jp", b"\x8f\xb0\xa1")[0][3] == "丂"                  # three bytes: jis0212
    assert one("euc-jp", b"\x8f\xb0AB")[0][0] == 2 and one("euc-jp", b"\x8f\xb0AB")[0][3] == "AB"  # 3rd byte ASCII: read again
    assert one("euc-jp", b"\x8f\xb0\x80AB")[0][0] == 3                  # 3rd byte invalid, not ASCII: consumed
    assert one("euc-jp", b"\x8f\x8eAB")[0][0] == 2                      # 8F + non-A1..FE: error, both consumed
    assert one("euc-jp", b"\x8fAB")[0][0] == 1
    assert one("euc-jp", b"\xa4\x8eAB")[0][0] == 2                      # lead + invalid trail: consumed
    assert one("euc-jp", b"\x90AB")[0][0] == 1 and one("euc-jp", b"\xa0AB")[0][0] == 1 and one("euc-jp", b"\xffAB")[0][0] == 1
    assert one("euc-jp", b"\xad\xa1")[0][3] == "①" and one("euc-jp", b"\xf9\xa1")[0][3] == "纊"   # NEC row 13, IBM extension
    assert one("euc-jp", b"\xa1\xc0")[0][3] == "＼"                     # FF3C (Windows-31J flavour), not 005C
    assert [x[3] for x in one("euc-jp", b"AB\x8f\xb0")] == ["AB"]       # pending at the end with is_last


def test_oracle_shift_jis_and_euc_kr_follow_the_whatwg_algorithms_on_hand_derived_vectors():
    sj = lambda b: one("shift_jis", b)
    assert sj(b"\x82\xa0\x82\xa2")[0][3] == "あい"
    assert sj(b"\xb1\xdf")[0][3] == "ｱﾟ"                                # A1..DF: half-width katakana, one byte each
    assert sj(b"\x80AB")[0][3] == "\x80AB"                               # 0x80 is U+0080 (WHATWG; ICU calls it an error)
    assert sj(b"\xa0AB")[0][0] == 1 and sj(b"\xfdAB")[0][0] == 1 and sj(b"\xffAB")[0][0] == 1   # A0, FD..FF: one-byte errors
    assert sj(b"\x81\x7fAB")[0][0] == 1 and sj(b"\x81\x7fAB")[0][3] == "\x7fAB"   # 7F is no trail but ASCII: read again
    assert sj(b"\x81\xfdAB")[0][0] == 2                                  # FD is no trail: both consumed
    assert sj(b"\xf0\x40")[0][3] == "\ue000" and sj(b"\xf9\xfc")[0][3] == "\ue757"   # user-defined pointers 8836..10715 -> U+E000..
    assert sj(b"\xfa\x40")[0][3] == "ⅰ" and sj(b"\x87\x40")[0][3] == "①"         # IBM extension, NEC row 13
    assert sj(b"\x81\x5f")[0][3] == "＼"                                  # trail 5C..7E are ASCII letters that belong to the pair
    assert [x[3] for x in sj(b"AB\x82")] == ["AB"]
    kr = lambda b: one("euc-kr", b)
    assert kr(b"\xb0\xa1\xb3\xaa")[0][3] == "가나"
    assert kr(b"\x81\x41")[0][3] == "갂"                                  # UHC extension: trail from 0x41
    assert kr(b"\x81\x40AB")[0][0] == 1 and kr(b"\x81\x40AB")[0][3] == "@AB"     # 0x40 is below the trail range: ASCII, read again
    assert kr(b"\xb0\xffAB")[0][0] == 2 and kr(b"\x80AB")[0][0] == 1 and kr(b"\xffAB")[0][0] == 1
    assert kr(b"\xc9\xa1AB")[0][0] == 2                                  # user-defined row: not in the index


def test_replacement_never_prints():
    ms = rc.missions(encodings=["replacement", "utf-8"], chars_min="3")
    data = b"plain text \x1b$)C and more text\n" * 2000
    want = sxo.run_cli(ms, [data], radix="x")
    assert b"replacement" not in want and want.count(b"UTF-8") > 1000
    assert run_cli_product(ms, [data], radix="x") == want
    assert run_cli_product(ms, [data], radix="x", chunk_bytes=4096) == want


# ---- the product's host side against the oracle ---------------------------------------------------------------------

def soup(enc, rng, n):
    txt, codec = TEXT[enc], CODEC[enc]
    nasty = [0x8E, 0x8F, 0xA1, 0xFE, 0x81, 0x80, 0xFF, 0x40, 0x7E, 0xA4, 0x88, 0x62, 0xA5, 0x0A, 0x20, 0xB0, 0x9F, 0xE0, 0xFC, 0xFD, 0xA0, 0xDF, 0x7F, 0x41]
    if enc in ("gbk", "gb18030"):   # digits: second and fourth bytes of the four-byte tokens
        nasty += [0x30, 0x39, 0x35, 0x81, 0x30, 0x84, 0x31, 0x90, 0x32, 0xE3, 0x39]
    out = bytearray()
    while len(out) < n:
        r = rng.random()
        if r < 0.3: out += txt[rng.randrange(len(txt)):][:rng.randrange(1, 40)].encode(codec, "ignore")
        elif r < 0.5: out += rng.randbytes(rng.randrange(1, 100))
        elif r < 0.6: out += b"plain ascii text %d " % rng.randrange(1000)
        elif r < 0.72: out += bytes(rng.choice(nasty) for _ in range(rng.randrange(1, 30)))
        elif r < 0.8: out += b"\x00" * rng.randrange(1, 300)
        elif r < 0.85: out += txt.encode(codec, "ignore")[rng.randrange(2):] * rng.randrange(1, 12)   # long stretches without any ASCII
        else: out += txt.encode(codec, "ignore") * rng.randrange(1, 4)
    return bytes(out[:n])


DBCS_FLAGS = [
    dict(chars_min="4", unicode_block_filter=ALL),
    dict(chars_min="3", output_line_len="16", unicode_block_filter="Cjk"),
    dict(chars_min="2", unicode_block_filter="Asian", same_unicode_block=True),
    dict(chars_min="5", grep_char="0x20", unicode_block_filter=ALL),
    dict(chars_min="10", unicode_block_filter="Kana"),
    dict(chars_min="1", output_line_len="6", unicode_block_filter="0x1008", ascii_filter="None"),   # C3 (Latin-1) yes, CC (combining) no: the Big5 pairs
    dict(chars_min="4", unicode_block_filter="Common"),  # no CJK at all: only ASCII and what maps below U+0800
]


@pytest.mark.parametrize("enc", ENCS)
@pytest.mark.parametrize("flags", DBCS_FLAGS, ids=lambda f: "n" + f["chars_min"] + "-" + f["unicode_block_filter"])
def test_host_replay_equals_oracle(enc, flags):
    rng = random.Random(zlib.crc32((enc + repr(sorted(flags.items()))).encode()))
    data = soup(enc, rng, 150_000)
    ms = rc.missions(encodings=[enc, "utf-8"], **flags)
    want = sxo.run_cli(ms, [data], radix="x")
    assert len(want) > 200
    assert run_cli_product(ms, [data], radix="x") == want
    for chunk in (4096, 8192, 65536):
        assert run_cli_product(ms, [data], radix="x", chunk_bytes=chunk) == want, chunk


@pytest.mark.parametrize("enc", ENCS)
def test_slice_start_probe_compares_two_characters(enc):
    """finding_collection.rs:176-207: a slice whose first character is completed from a lead byte of the slice before is
    marked `<` when a fresh decoder yields something else for the slice's first 8 output bytes.  The first character can
    coincide — lead L pending, slice = L L B ...: the true tokens are (L L)(L B), a fresh decoder sees (L L)(B L) — so
    the second one decides, and the 8-byte probe buffer must hold two characters for every decoder of this family
    (found by tools/gpu_fuzz.py: the product's two-byte decoder wanted 8 free bytes per character, the oracle's 4)."""
    ms = rc.missions(encodings=[enc], chars_min="2", unicode_block_filter=ALL, ascii_filter="All-Ctrl")
    for L, B in ((0xA4, 0xED), (0xB0, 0xA1), (0x88, 0x62), (0xE0, 0x9F)):
        body = bytes([L, L, B, L, B, L, B, L, B, L, B]) + b" and so on\n"
        for pad in (4095, 4094):   # pad 4094: the grid falls the other way
            data = b"x" * pad + bytes([L]) + body + b"\x00" * 64
            want = sxo.run_cli(ms, [data], radix="x")
            for chunk in (None, 4096):
                assert run_cli_product(ms, [data], radix="x", chunk_bytes=chunk) == want, (hex(L), hex(B), pad, chunk)


@pytest.mark.parametrize("enc", ENCS)
def test_tokens_across_chunk_and_file_boundaries(enc):
    """A token cut by a chunk boundary (lead | trail, 8F | xx | xx) is finished from the carried decoder; the token
    grid of the next chunk starts behind it; the state is carried over file boundaries as in the reference."""
    codec = CODEC[enc]
    word = TEXT[enc][:24].encode(codec, "ignore")
    ms = rc.missions(encodings=[enc], chars_min="3", unicode_block_filter=ALL)
    for shift in range(0, 7):
        for extra in (b"", b"\x8f\xb0\xa1\x8f\xb0\xa1" if enc == "euc-jp" else b"\x88\x62\x88\xa5"):
            lead_byte = b"\x88" if enc == "shift_jis" else b"\xa4"
            data = b"\x00" * (4096 - 5 - shift) + extra + word * 3 + b"\x00" * 100
            data += lead_byte * (8192 - len(data) % 8192 - 1 - shift) + word + b"\n" * 50   # a long stretch of lead-range bytes over a boundary
            want = sxo.run_cli(ms, [data], radix="x")
            assert run_cli_product(ms, [data], radix="x", chunk_bytes=4096) == want, (shift, extra)
            # two files: the pending lead byte survives the end of the first one
            cut = 4096 + 1
            want2 = sxo.run_cli(ms, [data[:cut], data[cut:]], radix="x")
            assert run_cli_product(ms, [data[:cut], data[cut:]], radix="x") == want2, (shift, extra)


@pytest.mark.parametrize("enc", ENCS)
@pytest.mark.parametrize("flags", DBCS_FLAGS[:5], ids=lambda f: "n" + f["chars_min"] + "-" + f["unicode_block_filter"])
def test_device_replay_core_emulated_on_cpu_equals_oracle(enc, flags):
    """The device's stage B (sx_replay_core.hpp compiled for the host, driven like the kernels drive it: one
    region per head run from derived state, with and without the shortcuts) for the double-byte decoders."""
    import test_replay_core as trc
    from test_sharded_gloo import oracle_findings
    core = trc.load_core()
    rng = random.Random(zlib.crc32((enc + "emu" + repr(sorted(flags.items()))).encode()))
    m = rc.missions(encodings=[enc], **flags)[0]
    long_run = max(1, min(m["chars_min_nb"], m["output_line_char_nb_max"]))
    done = 0
    for data in (soup(enc, rng, 60_000), soup(enc, rng, 30_001), rng.randbytes(40_000)):
        runs = sxo.runs(m, data, min_chars=long_run)
        want = [(p, pr, s, c, si) for p, pr, s, c, _, si in oracle_findings([m], data)]
        for skip in (1, 0):
            got = trc.emulate_device_stage_b(core, m, data, runs, skip=skip)
            if got is None:
                continue
            assert got == want, (skip, next(((a, b) for a, b in zip(got, want) if a != b), (len(got), len(want))))
            done += 1
    assert done > 0 or "grep_char" in flags   # (with -g a region can exceed the harness's 64 windows: given back by design)


@pytest.mark.parametrize("enc", ["gbk", "gb18030"])
def test_gb18030_digit_read_again_is_a_character_of_the_next_window(enc):
    """lead + digit as the last two bytes of a window, the token ends in an error in the next one: the digit is decoded again
    and delivered there (position = that window's start), although its byte — and so its run — lies in the window before and
    no run crosses the edge.  The region must go on while lead + digit are pending (found by tools/gpu_fuzz.py, seed 6868)."""
    import test_replay_core as trc
    from test_sharded_gloo import oracle_findings
    core = trc.load_core()
    flags = dict(chars_min="1", output_line_len="8", ascii_filter="0x7ffffffe000000007ffffffe00000000", unicode_block_filter="Latin")
    ms = rc.missions(encodings=[enc], **flags)
    body = bytes.fromhex("00000000000000be49bbba3317c18aef38" "eb4ab4f6ac9cc7fde6cbf71e1dc3e3b65cb2bf3f872b17a7e8ccc4e0ba6840f6")
    for pad in (15, 31, 4079):   # the window edge right behind the digit; the last one: a slice edge too
        data = b"\x00" * pad + body + b"\x00" * 200
        want = sxo.run_cli(ms, [data], radix="x")
        assert b"\t8\n" in want
        for chunk in (None, 4096):
            assert run_cli_product(ms, [data], radix="x", chunk_bytes=chunk) == want, (pad, chunk)
        runs = sxo.runs(ms[0], data, min_chars=1)
        wantf = [(p, pr, s, c, si) for p, pr, s, c, _, si in oracle_findings([ms[0]], data)]
        for skip in (1, 0):
            assert trc.emulate_device_stage_b(core, ms[0], data, runs, skip=skip) == wantf, (pad, skip)


def text_lines(enc, rng, n_lines):
    txt = TEXT[enc]
    lines = []
    for _ in range(n_lines):
        a = rng.randrange(len(txt) - 5)
        lines.append((txt[a:a + rng.randrange(3, 70)] + rng.choice(["", " 123", "abc", "9"])).encode(CODEC[enc], "ignore")
                     + rng.choice([b"\n", b"\r\n", b"\x00", b"\n\n"]))
    return b"".join(lines)


@pytest.mark.parametrize("enc", ["gbk", "gb18030", "big5", "euc-kr"])
def test_text_lines_cut_into_pieces_equal_oracle(enc):
    """Lines of CJK text (each crosses several window starts) framed by line feeds: the runs are cut into pieces at the window
    starts (for gb18030 / GBK only runs that verify as exact: sx_replay_core.hpp gb_run_is_exact) — host replay with the
    pieces the device would hand over, and the host-compiled device core driven as the device drives it."""
    import test_replay_core as trc
    from test_sharded_gloo import oracle_findings
    core = trc.load_core()
    data = text_lines(enc, random.Random(zlib.crc32(enc.encode())), 400)
    for flags in (dict(chars_min="4", unicode_block_filter=ALL), dict(chars_min="2", output_line_len="12", unicode_block_filter="Cjk")):
        ms = rc.missions(encodings=[enc], **flags)
        want = sxo.run_cli(ms, [data], radix="x")
        assert want.count(b"\n") > 300
        for chunk in (None, 4096):
            assert run_cli_product(ms, [data], radix="x", chunk_bytes=chunk, pieces=True) == want, (flags, chunk)
        m = ms[0]
        runs = sxo.runs(m, data, min_chars=max(1, min(m["chars_min_nb"], m["output_line_char_nb_max"])))
        cut = trc.split_runs(core, m, data, runs)
        assert len(cut) > len(runs)      # pieces were made
        wantf = [(p, pr, s, c, si) for p, pr, s, c, _, si in oracle_findings([m], data)]
        for skip in (1, 0):
            assert trc.emulate_device_stage_b(core, m, data, runs, skip=skip) == wantf, (flags, skip)
