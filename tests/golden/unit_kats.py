"""Golden vectors: the known answers of the reference's in-file unit tests,
transcribed as data (inputs and expected outputs only).

Sources (/root/reference/src): scanner.rs:192-559, finding_collection.rs:430-502,
main.rs:233-305, helper.rs:478-809.  Byte literals are the reference's test
inputs; expectations are its assert_eq! values.  Transcribed by hand; no
script (the reference is Rust and cannot run here).
"""
from refconfig import (AF_ALL, AF_CTRL, AF_WHITESPACE, UBF_ALL, UBF_GREEK, UBF_INVALID, UBF_LATIN, UBF_ACCENTS,
                       UBF_NONE, mission)

X_USER_DEFINED, UTF8 = 0, 1

FILTER_ALL_VALID = dict(af=AF_ALL, ubf=UBF_ALL & ~UBF_INVALID, grep_char=None)       # mission.rs:55-59
FILTER_LATIN = dict(af=AF_ALL & ~AF_CTRL | AF_WHITESPACE, ubf=UBF_LATIN | UBF_ACCENTS, grep_char=None)  # :64-68

# scanner.rs:105-191
MISSION_ALL_UTF8 = mission(encoding=UTF8, counter_offset=10_000, chars_min_nb=3, output_line_char_nb_max=10,
                           **FILTER_ALL_VALID)
MISSION_LATIN_UTF8 = mission(encoding=UTF8, counter_offset=10_000, chars_min_nb=3, output_line_char_nb_max=10,
                             **FILTER_LATIN)
MISSION_LATIN_UTF8_GREP42 = mission(encoding=UTF8, counter_offset=10_000, chars_min_nb=3,
                                    output_line_char_nb_max=10, af=AF_ALL & ~AF_CTRL | AF_WHITESPACE,
                                    ubf=UBF_LATIN, grep_char=42)
MISSION_ALL_X_USER_DEFINED = mission(encoding=X_USER_DEFINED, counter_offset=10_000, chars_min_nb=3,
                                     output_line_char_nb_max=10, **FILTER_ALL_VALID)
MISSION_ASCII = mission(encoding=X_USER_DEFINED, counter_offset=10_000, chars_min_nb=3,
                        output_line_char_nb_max=10, af=AF_ALL & ~AF_CTRL | AF_WHITESPACE, ubf=UBF_NONE,
                        grep_char=None)
MISSION_REAL_DATA_SCAN = mission(encoding=UTF8, counter_offset=10_000, chars_min_nb=4,
                                 output_line_char_nb_max=60, **FILTER_LATIN)


def F(position, precision, s):
    return dict(position=position, precision=precision, s=s)


# Each scenario: mission + ordered calls of FindingCollection::from.
# A call: input, is_last, then expectations (any subset):
#   findings (exact list) | findings_prefix (first k) | n_findings | n_findings_not
#   first_byte_position, consumed, maybe_cut, leftover, arena_prefix
SCAN_KATS = [
    dict(name="scan_input_buffer_chunks", src="scanner.rs:192-221", mission=MISSION_ALL_UTF8, calls=[
        dict(input=b"a234567890b234567890c234", is_last=True,
             findings=[F(10000, "Exact", "a234567890"), F(10000, "After", "b234567890"), F(10020, "Exact", "c234")],
             maybe_cut=False, first_byte_position=10000, consumed=10024),
    ]),
    dict(name="scan_store_in_scanner_state", src="scanner.rs:223-255", mission=MISSION_ALL_UTF8, calls=[
        dict(input=b"a234567890b234567890c2", is_last=True,
             findings=[F(10000, "Exact", "a234567890"), F(10000, "After", "b234567890"), F(10020, "Exact", "c2")],
             maybe_cut=False, first_byte_position=10000, consumed=10022),
    ]),
    dict(name="split_str_iterator_and_store_in_scanner_state", src="scanner.rs:257-304",
         mission=MISSION_ALL_UTF8, calls=[
        dict(input=b"You\xC0\x82\xC0co", is_last=False, findings=[F(10000, "Exact", "You")], leftover="co",
             first_byte_position=10000, consumed=10008),
        dict(input=b"me\xC0\x82\xC0home.", is_last=True,
             findings=[F(10008, "Before", "come"), F(10013, "Exact", "home.")], leftover="",
             first_byte_position=10008, consumed=10018),
    ]),
    dict(name="grep_in_scan", src="scanner.rs:306-350", mission=MISSION_LATIN_UTF8_GREP42, calls=[
        dict(input=b"You\xC0\x82\xC0co", is_last=False, findings=[], leftover="co", first_byte_position=10000,
             consumed=10008),
        dict(input=b"me*\xC0\x82\xC0ho*me.\x82", is_last=True,
             findings=[F(10008, "Before", "come*"), F(10014, "Exact", "ho*me.")], leftover="",
             first_byte_position=10008, consumed=10021),
    ]),
    dict(name="scan_buffer_split_multibyte", src="scanner.rs:352-412", mission=MISSION_ALL_UTF8, calls=[
        dict(input=b"word\xe2\x82", is_last=False),
        dict(input=b"\xacoh\xC0no no", is_last=False, findings_prefix=[F(10006, "Before", "word€oh")],
             first_byte_position=10006, consumed=10015),
        dict(input=b"\xe2\x82\xacStream end.", is_last=True,
             findings=[F(10015, "Before", "no no€Stre"), F(10015, "After", "am end.")],
             first_byte_position=10015, consumed=10029),
    ]),
    dict(name="to_short1", src="scanner.rs:414-470", mission=MISSION_ALL_UTF8, calls=[
        dict(input=b"ii\xC0abc\xC0\xC1de\xC0fgh\xC0ijk", is_last=False,
             findings=[F(10003, "Exact", "abc"), F(10011, "Exact", "fgh")], first_byte_position=10000,
             consumed=10018, maybe_cut=False, leftover="ijk"),
        dict(input=b"b\xC0\x82c\xC0def", is_last=True,
             findings=[F(10018, "Before", "ijkb"), F(10023, "Exact", "def")], first_byte_position=10018,
             consumed=10026, maybe_cut=False, leftover=""),
    ]),
    dict(name="to_short2", src="scanner.rs:472-531", mission=MISSION_LATIN_UTF8, calls=[
        dict(input="ii€ääà€€de€fgh€ijk".encode(), is_last=False,
             findings=[F(10000, "Exact", "ääà"), F(10020, "Before", "fgh")],
             first_byte_position=10000, consumed=10031, maybe_cut=False, leftover="ijk"),
        dict(input=b"b\xC0\x82c\xC0def", is_last=True,
             findings=[F(10031, "Before", "ijkb"), F(10036, "Exact", "def")], first_byte_position=10031,
             consumed=10039, maybe_cut=False, leftover=""),
    ]),
    dict(name="field_with_zeros", src="scanner.rs:533-559", mission=MISSION_REAL_DATA_SCAN, calls=[
        dict(input=b"\x00\x00\x00\x00\x40\x00\x38\x00\x0c\x00\x40\x00\x2c\x00\x2b\x00", is_last=False,
             n_findings_not=1),
    ]),
    dict(name="ascii_emulation_all_valid", src="finding_collection.rs:430-465",
         mission=MISSION_ALL_X_USER_DEFINED, calls=[
        dict(input=b"abcdefg\x58\x59\x80\x82h\x83ijk\x89\x90", is_last=True,
             findings=[F(10000, "Exact", "abcdefgXY\uf780"), F(10000, "After", "\uf782h\uf783ijk\uf789\uf790")],
             arena_prefix=("abcdefgXY\uf780\uf782h\uf783ijk\uf789\uf790" + "\0" * 7).encode(),
             first_byte_position=10000, consumed=10018, maybe_cut=False, leftover=""),
    ]),
    dict(name="ascii_emulation_ascii_filter", src="finding_collection.rs:467-502", mission=MISSION_ASCII, calls=[
        dict(input=b"abcdefg\x58\x59\x80\x82h\x83ijk\x89\x90", is_last=False,
             findings=[F(10000, "Exact", "abcdefgXY"), F(10000, "After", "ijk")],
             arena_prefix=("abcdefgXY\uf780\uf782h\uf783ijk\uf789\uf790" + "\0" * 7).encode(),
             first_byte_position=10000, consumed=10018, maybe_cut=False, leftover=""),
    ]),
]

# main.rs:198-305 — two missions over the same buffer, then kmerge.
MERGER_KAT = dict(
    src="main.rs:233-305",
    flags=dict(encodings=["ascii", "utf-8"], chars_min="5", same_unicode_block=True, output_line_len="30",
               counter_offset="5000"),
    input="abcdefgÜhijklmn€opÜqrstuvwÜxyz".encode(),
    per_mission=[["abcdefg", "hijklmn", "qrstuvw"], ["abcdefgÜhijklmn", "opÜqrstuvwÜxyz"]],
    merged=[("abcdefg", 5000, "Exact", 0), ("hijklmn", 5000, "After", 0), ("qrstuvw", 5000, "After", 0),
            ("abcdefgÜhijklmn", 5000, "Exact", 1), ("opÜqrstuvwÜxyz", 5000, "After", 1)],
)


def S(s, **flags):
    d = dict(s=s)
    d.update(flags)
    return d


_LAT = dict(af=AF_ALL, ubf=UBF_LATIN)
_LATGR = dict(af=AF_ALL, ubf=UBF_LATIN | UBF_GREEK)
_ASC = dict(af=AF_ALL, ubf=UBF_NONE)
E = "€"

# helper.rs:478-809.  args: (chars_min_nb, same_block, last_s_was_maybe_cut, invalid_bytes_after, grep, q)
# q=None means inp.len() in bytes, as the tests pass `b.len()`.
SPLIT_KATS = [
    dict(src="helper.rs:487-499", filt=_LAT, inp=f"{E}abc{E}defg{E}hijk{E}lm{E}opq", args=(3, False, False, False, None, None),
         out=[S("abc", completes=False), S("defg"), S("hijk"), S("opq")]),
    dict(src="helper.rs:501-520", filt=_LAT, inp=f"ab{E}{E}defg{E}hijk{E}lm{E}opq", args=(3, False, True, False, None, None),
         out=[S("ab", completes=True, min_ok=False, again=False), S("defg"), S("hijk"),
              S("opq", maybe_cut=True, min_ok=True, again=True)]),
    dict(src="helper.rs:522-535", filt=_LAT, inp=f"ab{E}{E}defg{E}hijk{E}lm{E}op", args=(3, False, False, False, None, None),
         out=[S("defg", completes=False), S("hijk"), S("op", maybe_cut=True, min_ok=False, again=True)]),
    dict(src="helper.rs:537-550", filt=_LAT, inp=f"{E}abc{E}defg{E}hijk{E}lm", args=(4, False, False, False, None, None),
         out=[S("defg"), S("hijk", maybe_cut=False), S("lm", maybe_cut=True, min_ok=False, again=True)]),
    dict(src="helper.rs:552-564", filt=_LAT, inp=f"{E}abc{E}defg{E}hijk{E}lmno{E}", args=(4, False, False, False, None, None),
         out=[S("defg"), S("hijk"), S("lmno", maybe_cut=False, min_ok=True, again=False)]),
    dict(src="helper.rs:566-593", filt=_LAT, inp=f"abc{E}defghiÜjklmnpqrs{E}", args=(4, False, False, False, None, 7),
         out=[S("defghiÜ", completes=False, maybe_cut=True, again=False, min_ok=True),
              S("jklmnpq", completes=True, maybe_cut=True, again=False, min_ok=True),
              S("rs", completes=True, maybe_cut=False, again=False, min_ok=False)]),
    dict(src="helper.rs:595-604", filt=_LAT, inp="abcdefghijklm", args=(4, False, False, False, None, None),
         out=[S("abcdefghijklm", completes=False, maybe_cut=True, again=False, min_ok=True)]),
    dict(src="helper.rs:606-615", filt=_LAT, inp=f"abcdefghijklm{E}", args=(4, False, False, False, None, None),
         out=[S("abcdefghijklm", completes=False, maybe_cut=False, again=False, min_ok=True)]),
    dict(src="helper.rs:617-626", filt=_LAT, inp=f"öö{E}{E}ääää{E}üü{E}éééé{E}",
         args=(4, False, True, False, None, None), out=[S("öö"), S("ääää"), S("éééé")]),
    dict(src="helper.rs:631-640", filt=_ASC, inp=f"öö{E}{E}ääää{E}üü{E}éééé{E}",
         args=(4, False, True, False, None, None), out=[]),
    dict(src="helper.rs:652-660", filt=_LATGR, inp=f"0α1βγöäü{E}α2βγöäüöαβγαg34αäβüäöüαβγöäü",
         args=(3, False, False, False, None, None),
         out=[S("0α1βγöäü"),
              S("α2βγöäüöαβγαg34αäβüäöüαβγöäü")]),
    dict(src="helper.rs:662-676", filt=_LATGR, inp=f"0α1βγöäü{E}α2βγöäüöαβγαg34αäβüäöü",
         args=(4, True, False, False, None, None),
         out=[S("0α1βγ"), S("α2βγ"), S("öäüö"),
              S("αβγαg34α"), S("üäöü")]),
    dict(src="helper.rs:688-707", filt=_LAT, inp=f"ac{E}{E}xefg{E}xijk{E}xm{E}xp", args=(3, False, True, False, None, None),
         out=[S("ac", completes=True, again=False, maybe_cut=False), S("xefg"), S("xijk"),
              S("xp", completes=False, again=True, maybe_cut=True)]),
    dict(src="helper.rs:709-727", filt=_LAT, inp=f"ac{E}{E}xefg{E}xijk{E}xm{E}xp", args=(2, False, True, False, ord("b"), 3),
         out=[S("ac", completes=True, again=False, maybe_cut=False)]),
    dict(src="helper.rs:729-786", filt=_LAT, inp=f"ac{E}{E}xefg{E}xijk{E}xm{E}xp", args=(2, False, True, False, ord("x"), 3),
         out=[S("ac", completes=True, again=False, maybe_cut=False, grep_ok=False),
              S("xef", completes=False, again=False, maybe_cut=True, grep_ok=True),
              S("g", completes=True, again=False, maybe_cut=False, grep_ok=False),
              S("xij", completes=False, again=False, maybe_cut=True, grep_ok=True),
              S("k", completes=True, again=False, maybe_cut=False, grep_ok=False),
              S("xm", completes=False, again=False, maybe_cut=False, grep_ok=True),
              S("xp", completes=False, again=True, maybe_cut=True, grep_ok=True)]),
    dict(src="helper.rs:788-808", filt=_LAT, inp=f"öä{E}{E}äüöä{E}äüöö{E}üö{E}üü",
         args=(3, False, False, False, ord("y"), None), out=[S("üü", completes=False, again=True, maybe_cut=True)]),
]
