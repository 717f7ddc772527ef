"""Test helper: the filter constants and the Mission resolution rules of the
reference (src/mission.rs:32-161,225-253 constants; :167-218,255-274 alias
tables; :448-504 parsers; :514-703 Missions::new; :713-749 parse_enc_opt;
src/options.rs:12-33 defaults).  Produces plain dicts that the ctypes bindings
of the oracle (sxo_binding) and of the product (stringsext_amd) turn into
their own mission structs."""

UBF_ALL = 0xFFFF_FFFF_FFFF_FFFF
UBF_NONE = 0
UBF_INVALID = 0xFFE0_0000_0000_0003
UBF_LATIN = 0x01FC
UBF_ACCENTS = 0x3000
UBF_GREEK = 0xC000
UBF_IPA = 0x0700
UBF_CYRILLIC = 0x001F_0000
UBF_ARMENIAN = 0x0020_0000
UBF_HEBREW = 0x00C0_0000
UBF_ARABIC = 0x2F00_0000
UBF_SYRIAC = 0x1000_0000
UBF_AFRICAN = 0xFFE0_0000
UBF_COMMON = 0xFFFF_FFFC
UBF_KANA = 0x0000_0008_0000_0000
UBF_CJK = 0x0000_03F0_0000_0000
UBF_HANGUL = 0x0000_3800_0000_0000
UBF_ASIAN = 0x0000_3FFC_0000_0000
UBF_PUA = 0x0010_4000_0000_0000
UBF_MISC = 0x0000_8006_0000_0000
UBF_UNCOMMON = 0x000F_0000_0000_0000
UBF_ALL_VALID = UBF_ALL & ~UBF_INVALID

AF_ALL = 0xFFFF_FFFF_FFFF_FFFF_FFFF_FFFF_FFFF_FFFE
AF_NONE = 0
AF_CTRL = 0x8000_0000_0000_0000_0000_0000_FFFF_FFFF
AF_WHITESPACE = 0x0000_0000_0000_0000_0000_0001_0000_1E00
AF_DEFAULT = AF_ALL & ~AF_CTRL

# order matters: first prefix match wins (mission.rs:486-491)
UBF_ALIASES = [
    ("African", UBF_AFRICAN), ("All-Asian", UBF_ALL & ~UBF_INVALID & ~UBF_ASIAN),
    ("All", UBF_ALL & ~UBF_INVALID), ("Arabic", UBF_ARABIC | UBF_SYRIAC), ("Armenian", UBF_ARMENIAN),
    ("Asian", UBF_ASIAN), ("Cjk", UBF_CJK), ("Common", UBF_COMMON), ("Cyrillic", UBF_CYRILLIC),
    ("Default", UBF_ALL & ~UBF_INVALID), ("Greek", UBF_GREEK), ("Hangul", UBF_HANGUL),
    ("Hebrew", UBF_HEBREW), ("Kana", UBF_KANA), ("Latin", UBF_LATIN | UBF_ACCENTS),
    ("None", (~UBF_ALL) & UBF_ALL), ("Private", UBF_PUA), ("Uncommon", UBF_UNCOMMON | UBF_PUA),
]
AF_ALIASES = [
    ("All", AF_ALL), ("All-Ctrl", AF_ALL & ~AF_CTRL), ("All-Ctrl+Wsp", AF_ALL & ~AF_CTRL | AF_WHITESPACE),
    ("Default", AF_DEFAULT), ("None", AF_NONE), ("Wsp", AF_WHITESPACE),
]

ENC_IDS = {"x-user-defined": 0, "utf-8": 1, "utf-16le": 2, "utf-16be": 3, "koi8-r": 16, "ibm866": 17,
           "iso-8859-2": 18, "iso-8859-5": 19, "iso-8859-15": 20, "windows-1251": 21, "windows-1252": 22,
           "iso-8859-3": 23, "iso-8859-4": 24, "iso-8859-6": 25, "iso-8859-7": 26, "iso-8859-8": 27,
           "iso-8859-8-i": 28, "iso-8859-10": 29, "iso-8859-13": 30, "iso-8859-14": 31, "iso-8859-16": 32,
           "koi8-u": 33, "macintosh": 34, "windows-874": 35, "windows-1250": 36, "windows-1253": 37,
           "windows-1254": 38, "windows-1255": 39, "windows-1256": 40, "windows-1257": 41, "windows-1258": 42,
           "x-mac-cyrillic": 43, "big5": 64, "euc-jp": 65, "shift_jis": 66, "euc-kr": 67, "gb18030": 68, "gbk": 69, "replacement": 70,
           "iso-2022-jp": 71}


def _parse_int(s):
    if s is None or s == "":
        return None
    t = s.strip()
    return int(t[2:], 16) if t[:2] == "0x" else int(t)


def _parse_filter(s, aliases):
    if s is None:
        return None
    t = s.strip()
    if len(t) >= 2 and t[:2] == "0x":
        return int(t[2:], 16)
    if s == "":
        return None
    for name, val in aliases:
        padded = name.ljust(12)
        if len(t) <= len(padded) and padded.startswith(t):
            return val
    raise ValueError(f"filter name `{t}` is not valid")


def missions(encodings=(), chars_min=None, same_unicode_block=False, ascii_filter=None,
             unicode_block_filter=None, grep_char=None, output_line_len=None, counter_offset=None):
    """Missions::new (mission.rs:514-703)."""
    f_off = _parse_int(counter_offset) if isinstance(counter_offset, str) else counter_offset
    f_min = _parse_int(chars_min) if isinstance(chars_min, str) else chars_min
    f_af = _parse_filter(ascii_filter, AF_ALIASES)
    f_ubf = _parse_filter(unicode_block_filter, UBF_ALIASES)
    f_grep = _parse_int(grep_char) if isinstance(grep_char, str) else grep_char
    f_q = _parse_int(output_line_len) if isinstance(output_line_len, str) else output_line_len
    out = []
    for mid, opt in enumerate(list(encodings) or ["UTF-8"]):
        parts = opt.split(",")
        while parts and parts[-1] == "" and len(parts) > 1:  # split_terminator
            parts.pop()
        parts += [None] * (5 - len(parts))
        name = parts[0].strip() if parts[0] else "UTF-8"
        n = _parse_int(parts[1])
        af = _parse_filter(parts[2], AF_ALIASES)
        ubf = _parse_filter(parts[3], UBF_ALIASES)
        grep = _parse_int(parts[4])
        is_ascii = name == "ascii"
        n = n if n is not None else (f_min if f_min is not None else 4)
        af = af if af is not None else (f_af if f_af is not None else AF_DEFAULT)
        ubf = ubf if ubf is not None else (f_ubf if f_ubf is not None else (UBF_NONE if is_ascii else UBF_COMMON))
        grep = grep if grep is not None else f_grep
        enc_name = "x-user-defined" if is_ascii else name.lower()
        out.append(dict(mission_id=mid, encoding=ENC_IDS[enc_name], chars_min_nb=n,
                        require_same_unicode_block=bool(same_unicode_block), grep_char=grep, af=af, ubf=ubf,
                        output_line_char_nb_max=f_q if f_q is not None else 64,
                        counter_offset=f_off if f_off is not None else 0, print_encoding_as_ascii=is_ascii))
    return out


def mission(**kw):
    """One mission dict with test-style explicit fields."""
    d = dict(mission_id=0, encoding=1, chars_min_nb=4, require_same_unicode_block=False, grep_char=None,
             af=AF_DEFAULT, ubf=UBF_COMMON, output_line_char_nb_max=64, counter_offset=0,
             print_encoding_as_ascii=False)
    d.update(kw)
    return d
