"""The Mission front end behind the C-ABI (sx_missions_from_flags, sx_parse_enc_opt — SURVEY §8 f-2)
against the reference's own unit-test vectors (src/mission.rs:776-868, transcribed as data) and
against the Python restatement the other tests use (tests/refconfig.py).  No GPU needed."""
import os
import random

import pytest

import refconfig as rc
import stringsext_amd as sx

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
AF_DEFAULT, UBF_LATIN_ACC = rc.AF_DEFAULT, rc.UBF_LATIN | rc.UBF_ACCENTS

# src/mission.rs test_enc_opt_parser: (input, expected tuple) and the inputs that must fail
ENC_OPT_OK = [
    ("ascii", ("ascii", None, None, None, None)),
    ("utf-8,10,0x89AB,0xCDEF,0x2f", ("utf-8", 10, 0x89AB, 0xCDEF, 0x2F)),
    ("utf-8,10,0x89AB,0xCDEF,211", ("utf-8", 10, 0x89AB, 0xCDEF, 211)),
    (",,,,,", (None, None, None, None, None)),
    ("ascii,10,0x89AB", ("ascii", 10, 0x89AB, None, None)),
    ("ascii,10,Default", ("ascii", 10, AF_DEFAULT, None, None)),
    ("ascii,10,,Latin", ("ascii", 10, None, UBF_LATIN_ACC, None)),
]
ENC_OPT_ERR = ["ascii, 10n", "ascii,10,0x89,0x?B", "ascii,10,0x?9,0xAB", "ascii,1000000000000000000000,0x1,0x2",
               "ascii,10,0x1,0x2,0x3,0x4", "ascii,10,123", "ascii,10,,123", "ascii,10,my-no-encoding",
               "ascii,10,,my-no-encoding"]


@pytest.mark.parametrize("text,want", ENC_OPT_OK)
def test_parse_enc_opt_reference_vectors(text, want):
    assert sx.parse_enc_opt(text) == want


@pytest.mark.parametrize("text", ENC_OPT_ERR)
def test_parse_enc_opt_reference_errors(text):
    with pytest.raises(sx.SxError):
        sx.parse_enc_opt(text)


def test_baseline_configs_resolve_as_survey_8a_says():
    c1 = sx.missions_from_flags(encodings=["ascii"], chars_min="4")[0]
    assert (c1["encoding"], c1["print_encoding_as_ascii"], c1["af"], c1["ubf"], c1["chars_min_nb"]) == \
        (0, True, 0x7fffffff_ffffffff_ffffffff_00000000, 0, 4)
    c3 = sx.missions_from_flags(encodings=["utf-8", "utf-16le", "utf-16be"], chars_min="10", unicode_block_filter="African")
    assert [m["encoding"] for m in c3] == [1, 2, 3] and all(m["ubf"] == 0xffe0_0000 and m["output_line_char_nb_max"] == 64 for m in c3)
    c5 = sx.missions_from_flags(encodings=["utf-8,,,African", "koi8-r,,,Cyrillic"], chars_min="10")
    assert [(m["encoding"], m["ubf"]) for m in c5] == [(1, 0xffe0_0000), (16, 0x1f_0000)]
    assert sx.missions_from_flags()[0]["encoding"] == 1 and sx.missions_from_flags()[0]["chars_min_nb"] == 4   # UTF-8, n=4


def test_alias_prefix_quirks_and_labels():
    # the first alias the text is a prefix of wins: "All" is "All-Asian", "A" is "African" (mission.rs:486-491)
    m = sx.missions_from_flags(encodings=["utf-8"], unicode_block_filter="All", ascii_filter="All")[0]
    assert m["ubf"] == rc.UBF_ALL & ~rc.UBF_INVALID & ~rc.UBF_ASIAN and m["af"] == rc.AF_ALL
    assert sx.missions_from_flags(encodings=["utf-8,,,A"])[0]["ubf"] == rc.UBF_AFRICAN
    # WHATWG labels, case-insensitive, whitespace-tolerant; "ascii" is the reference's own pseudo-encoding
    for label, enc in [("UTF8", 1), (" utf-16 ", 2), ("unicodeFFFE", 3), ("latin1", 22), ("KOI8_R", 16), ("866", 17), ("cyrillic", 19),
                       ("l9", 20), ("x-user-defined", 0), ("windows-1251", 21), ("latin2", 18)]:
        assert sx.missions_from_flags(encodings=[label])[0]["encoding"] == enc, label
    # the rest of the WHATWG single-byte set, by some of their labels
    for label, name in [("latin3", "ISO-8859-3"), ("l4", "ISO-8859-4"), ("arabic", "ISO-8859-6"), ("ELOT_928", "ISO-8859-7"),
                        ("visual", "ISO-8859-8"), ("logical", "ISO-8859-8-I"), ("latin6", "ISO-8859-10"), ("iso885913", "ISO-8859-13"),
                        ("iso8859-14", "ISO-8859-14"), ("iso-8859-16", "ISO-8859-16"), ("koi8-ru", "KOI8-U"), ("mac", "macintosh"),
                        ("tis-620", "windows-874"), ("iso-8859-11", "windows-874"), ("cp1250", "windows-1250"), ("x-cp1253", "windows-1253"),
                        ("latin5", "windows-1254"), ("iso-8859-9", "windows-1254"), ("cp1255", "windows-1255"), ("windows-1256", "windows-1256"),
                        ("cp1257", "windows-1257"), ("x-cp1258", "windows-1258"), ("x-mac-ukrainian", "x-mac-cyrillic")]:
        enc = sx.missions_from_flags(encodings=[label])[0]["encoding"]
        assert sx.encoding_name(enc) == name and rc.ENC_IDS[name.lower()] == enc, label
    with pytest.raises(sx.SxError, match="invalid input encoding name"):
        sx.missions_from_flags(encodings=["utf-9"])
    # the legacy multi-byte encodings that are built in (help.rs:56-57; mission.rs:681)
    for label, name in [("big5", "Big5"), ("Big5-HKSCS", "Big5"), ("x-x-big5", "Big5"), ("EUC-JP", "EUC-JP"), ("x-euc-jp", "EUC-JP"),
                        ("sjis", "Shift_JIS"), ("windows-31j", "Shift_JIS"), ("MS_Kanji", "Shift_JIS"), ("korean", "EUC-KR"),
                        ("windows-949", "EUC-KR"), ("ks_c_5601-1987", "EUC-KR"), ("iso-2022-kr", "replacement"), ("hz-gb-2312", "replacement"),
                        ("iso-2022-jp", "ISO-2022-JP"), ("csISO2022JP", "ISO-2022-JP"),
                        ("gb18030", "gb18030"), ("GBK", "GBK"), ("gb2312", "GBK"), ("chinese", "GBK"), ("x-gbk", "GBK"), ("iso-ir-58", "GBK")]:
        enc = sx.missions_from_flags(encodings=[label])[0]["encoding"]
        assert sx.encoding_name(enc) == name and rc.ENC_IDS[name.lower()] == enc, label
    with pytest.raises(sx.SxError, match="ASCII codes < 128"):
        sx.missions_from_flags(encodings=["utf-8"], grep_char="200")
    with pytest.raises(sx.SxError, match="output-line-len"):
        sx.missions_from_flags(encodings=["utf-8"], output_line_len="5")
    with pytest.raises(sx.SxError, match="Too many items"):
        sx.missions_from_flags(encodings=["utf-8,1,0x1,0x2,3,4"])


def test_random_flag_sets_agree_with_the_python_restatement():
    rng = random.Random(99)
    encs = ["ascii", "utf-8", "UTF-16LE", "utf-16be", "koi8-r", "ibm866", "iso-8859-2", "iso-8859-5", "iso-8859-15",
            "windows-1251", "windows-1252", "x-user-defined", "", "iso-8859-7", "windows-1255", "koi8-u", "macintosh", "windows-874"]
    afs = [None, "", "All", "All-Ctrl", "All-Ctrl+Wsp", "Default", "None", "Wsp", "W", "0x7f", " 0xFFFF "]
    ubfs = [None, "", "African", "All-Asian", "All", "Arabic", "Armenian", "Asian", "Cjk", "Common", "Cyrillic", "Default", "Greek",
            "Hangul", "Hebrew", "Kana", "Latin", "None", "Private", "Uncommon", "C", "H", "0xfffc", "0x0"]
    nums = [None, "", "4", " 12 ", "0x10", "+7", "255"]
    for _ in range(400):
        e = []
        for _ in range(rng.randrange(0, 4)):
            parts = [rng.choice(encs)] + [rng.choice(nums) or "", rng.choice(afs) or "", rng.choice(ubfs) or "", rng.choice([None, "", "47", "0x2f"]) or ""]
            e.append(",".join(parts[:rng.randrange(1, 6)]))
        kw = dict(encodings=e, chars_min=rng.choice(nums), same_unicode_block=rng.random() < 0.3, ascii_filter=rng.choice(afs),
                  unicode_block_filter=rng.choice(ubfs), grep_char=rng.choice([None, "", "65", "0x41"]),
                  output_line_len=rng.choice([None, "", "6", "64", "0x20", "1000"]), counter_offset=rng.choice([None, "", "1500", "0x100"]))
        want = rc.missions(**kw)
        for w in want:   # the restatement keeps the reference's lower-case names; encoding ids are what the ABI carries
            pass
        got = sx.missions_from_flags(**kw)
        assert got == want, (kw, got, want)
