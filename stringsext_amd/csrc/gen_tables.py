#!/usr/bin/env python3
"""Generate sx_tables.inc — the PRODUCT's decoder tables, from CPython's codecs (the oracle's tables come from
ICU dumps instead, oracle/gen_tables.py; tests/test_tables.py compares the two cell by cell).

encoding_rs 0.8.34 is not vendored in /root/reference and there is no network, so the WHATWG index files are
not available.  CPython differs from them in known places, patched here:
  single byte
  * bytes 0x80..0x9F that CPython leaves undefined in the windows-125x / windows-874 code pages map to the C1
    control of the same value (WHATWG fills them); undefined bytes from 0xA0 up stay undefined in both;
  * windows-1255 0xCA = U+05BA; KOI8-U 0xAE = U+045E, 0xBE = U+040E (the WHATWG index is KOI8-RU's);
    x-mac-cyrillic 0xA2 = U+0490, 0xB6 = U+0491 (the WHATWG index is x-mac-ukrainian's); ISO-8859-8-I shares
    ISO-8859-8's index.
  Big5 (index-big5: Big5 + HKSCS-2008, Microsoft flavour in the standard rows)
  * `big5hkscs` everywhere, except that a cell `cp950` maps outside ETEN's block C6A1..C8FE takes cp950's value
    (12 symbols in rows A1/A2 and F9FE differ; A3E1 = U+20AC only cp950 has), and inside that block the six cells
    only cp950 has (C6CF, C6D3, C6D5, C6D7, C6DE, C6DF) are kept;
  * pointers 1133, 1135, 1164, 1166 (two code points each) are left 0: the decoder handles them.
  EUC-JP (index-jis0208 is Windows-31J's table: NEC row 13 and the IBM extension rows 89..92 included)
  * jis0208 from `cp932` through the Shift_JIS pointer arithmetic; jis0212 from `euc_jp`'s three-byte form with
    0xA2B7 = U+FF5E (CPython: U+007E).
  Shift_JIS: index jis0208 from `cp932` by Shift_JIS pointer (ICU agrees on every cell); EUC-KR: `cp949` (= windows-949,
  the WHATWG index; ICU has only the KS X 1001 part).
Parity of all legacy tables is UNPINNED (no reference test decodes any of them).

Order of the single-byte tables defines the encoding id (16 + index) and must only ever be appended to; names are
Encoding::name().
"""
import sys

TABLES = [
    ("KOI8-R", "koi8_r", False),
    ("IBM866", "cp866", False),
    ("ISO-8859-2", "iso8859_2", False),
    ("ISO-8859-5", "iso8859_5", False),
    ("ISO-8859-15", "iso8859_15", False),
    ("windows-1251", "cp1251", True),
    ("windows-1252", "cp1252", True),
    # --- the rest of the WHATWG single-byte set
    ("ISO-8859-3", "iso8859_3", False),
    ("ISO-8859-4", "iso8859_4", False),
    ("ISO-8859-6", "iso8859_6", False),
    ("ISO-8859-7", "iso8859_7", False),
    ("ISO-8859-8", "iso8859_8", False),
    ("ISO-8859-8-I", "iso8859_8", False),
    ("ISO-8859-10", "iso8859_10", False),
    ("ISO-8859-13", "iso8859_13", False),
    ("ISO-8859-14", "iso8859_14", False),
    ("ISO-8859-16", "iso8859_16", False),
    ("KOI8-U", "koi8_u", False),
    ("macintosh", "mac_roman", False),
    ("windows-874", "cp874", True),
    ("windows-1250", "cp1250", True),
    ("windows-1253", "cp1253", True),
    ("windows-1254", "cp1254", True),
    ("windows-1255", "cp1255", True),
    ("windows-1256", "cp1256", True),
    ("windows-1257", "cp1257", True),
    ("windows-1258", "cp1258", True),
    ("x-mac-cyrillic", "mac_cyrillic", False),
]

PATCHES = {
    "windows-1255": {0xCA: 0x05BA},
    "KOI8-U": {0xAE: 0x045E, 0xBE: 0x040E},
    "x-mac-cyrillic": {0xA2: 0x0490, 0xB6: 0x0491},
}


def table(codec, c1_fill, name=None):
    out = []
    for b in range(0x80, 0x100):
        try:
            cp = ord(bytes([b]).decode(codec))
        except UnicodeDecodeError:
            cp = b if (c1_fill and b < 0xA0) else 0
        cp = PATCHES.get(name, {}).get(b, cp)
        out.append(cp)
    return out


BIG5_N = 126 * 157
JIS_N = 94 * 94


def cps(codec, bs):
    try:
        return [ord(c) for c in bytes(bs).decode(codec)]
    except UnicodeDecodeError:
        return None


def big5_table():
    """pointer -> code point (0 = unmapped)"""
    t = [0] * BIG5_N
    for lead in range(0x81, 0xFF):
        for trail in list(range(0x40, 0x7F)) + list(range(0xA1, 0xFF)):
            ptr = (lead - 0x81) * 157 + (trail - (0x40 if trail < 0x7F else 0x62))
            in_eten_block = (lead == 0xC6 and trail >= 0xA1) or lead in (0xC7, 0xC8)
            ms = cps("cp950", [lead, trail])
            hk = cps("big5hkscs", [lead, trail])
            v = ms if (ms and not in_eten_block) else (hk or ms)
            if v and len(v) == 1:
                t[ptr] = v[0]
    return t


def jis0208_table():
    t = [0] * JIS_N
    for p in range(JIS_N):  # the index is shared with Shift_JIS: pointer -> Shift_JIS bytes -> cp932
        lead, trail = divmod(p, 188)
        b0 = lead + (0x81 if lead < 0x1F else 0xC1)
        b1 = trail + (0x40 if trail < 0x3F else 0x41)
        v = cps("cp932", [b0, b1])
        if v and len(v) == 1:
            t[p] = v[0]
    return t


def jis0212_table():
    t = [0] * JIS_N
    for a in range(0xA1, 0xFF):
        for b in range(0xA1, 0xFF):
            v = cps("euc_jp", [0x8F, a, b])
            if v and len(v) == 1:
                t[(a - 0xA1) * 94 + (b - 0xA1)] = v[0]
    t[(0xA2 - 0xA1) * 94 + (0xB7 - 0xA1)] = 0xFF5E
    return t


SJIS_N = 11280
EUCKR_N = 126 * 190


def shift_jis_table():
    """index jis0208 by Shift_JIS pointer, from cp932 (the user-defined pointers 8836..10715 stay 0: a rule in the decoder)"""
    t = [0] * SJIS_N
    for p in range(SJIS_N):
        if 8836 <= p <= 10715:
            continue
        lead, trail = divmod(p, 188)
        v = cps("cp932", [lead + (0x81 if lead < 0x1F else 0xC1), trail + (0x40 if trail < 0x3F else 0x41)])
        if v and len(v) == 1:
            t[p] = v[0]
    return t


def euc_kr_table():
    """the WHATWG euc-kr index is windows-949's: cp949"""
    t = [0] * EUCKR_N
    for lead in range(0x81, 0xFF):
        for trail in range(0x41, 0xFF):
            v = cps("cp949", [lead, trail])
            if v and len(v) == 1 and v[0] >= 0x80:
                t[(lead - 0x81) * 190 + (trail - 0x41)] = v[0]
    return t


GB_N = 126 * 190
GB_RANGE_LIMIT = 39420   # four-byte pointers below this are in the BMP ranges; 189000.. are U+10000..


def gb18030_tables():
    """index gb18030 (two-byte cells) and index gb18030 ranges as breakpoints (pointer, code point), from CPython's
    gb18030 codec, patched to the WHATWG indexes: 0xA8BC = U+1E3F with four-byte pointer 7457 = U+E7C7 (CPython has
    the GB18030-2000 assignment the other way round; ICU agrees with WHATWG), 0xA3A0 = U+3000 (both sources: U+E5E5)."""
    cells = [0] * GB_N
    for lead in range(0x81, 0xFF):
        for trail in list(range(0x40, 0x7F)) + list(range(0x80, 0xFF)):
            v = cps("gb18030", [lead, trail])
            if v and len(v) == 1:
                cells[(lead - 0x81) * 190 + (trail - (0x40 if trail < 0x7F else 0x41))] = v[0]
    cells[(0xA8 - 0x81) * 190 + (0xBC - 0x41)] = 0x1E3F
    cells[(0xA3 - 0x81) * 190 + (0xA0 - 0x41)] = 0x3000
    cp4 = []
    for p in range(GB_RANGE_LIMIT):
        v = cps("gb18030", [0x81 + p // 12600, 0x30 + (p // 1260) % 10, 0x81 + (p // 10) % 126, 0x30 + p % 10])
        assert v and len(v) == 1, p
        cp4.append(v[0])
    cp4[7457] = 0xE7C7
    ptrs, starts = [], []
    for p, cp in enumerate(cp4):
        if p == 0 or cp != cp4[p - 1] + 1:
            ptrs.append(p); starts.append(cp)
    return cells, ptrs, starts


def emit_array(fh, ctype, name, values, per_line, width):
    fh.write(f"static const {ctype} {name}[{len(values)}] = {{\n")
    for i in range(0, len(values), per_line):
        fh.write(" " + ",".join(f"0x{v:0{width}X}" for v in values[i:i + per_line]) + ",\n")
    fh.write("};\n")


def emit(prefix, fh):
    fh.write("/* GENERATED by gen_tables.py — do not edit. */\n")
    fh.write(f"#define {prefix.upper()}_N_SB_TABLES {len(TABLES)}\n")
    fh.write(f"static const char* const {prefix}_sb_names[{len(TABLES)}] = {{\n")
    for name, _, _ in TABLES:
        fh.write(f'    "{name}",\n')
    fh.write("};\n")
    fh.write(f"static const uint16_t {prefix}_sb_tables[{len(TABLES)}][128] = {{\n")
    for name, codec, fill in TABLES:
        t = table(codec, fill, name)
        fh.write(f"    /* {name} */ {{\n")
        for i in range(0, 128, 8):
            fh.write("        " + ", ".join(f"0x{v:04X}" for v in t[i:i + 8]) + ",\n")
        fh.write("    },\n")
    fh.write("};\n")
    # double-byte: Big5 as the low 16 bits + a bitmap of the cells in plane 2 (every astral Big5 character is)
    b5 = big5_table()
    assert all(v < 0x10000 or (v >> 16) == 2 for v in b5)
    plane2 = [0] * ((BIG5_N + 15) // 16)
    for p, v in enumerate(b5):
        if v >> 16:
            plane2[p >> 4] |= 1 << (p & 15)
    # one blob of uint16_t per encoding (it goes to the device as it is):
    #   Big5:   [BIG5_N low halves][BIG5_P2_WORDS bitmap words: pointer p is in plane 2 iff bit p&15 of word p>>4]
    #   EUC-JP: [JIS_N jis0208][JIS_N jis0212]
    fh.write(f"#define {prefix.upper()}_BIG5_N {BIG5_N}\n#define {prefix.upper()}_BIG5_P2_WORDS {len(plane2)}\n"
             f"#define {prefix.upper()}_JIS_N {JIS_N}\n")
    emit_array(fh, "uint16_t", f"{prefix}_big5", [v & 0xFFFF for v in b5] + plane2, 16, 4)
    emit_array(fh, "uint16_t", f"{prefix}_eucjp", jis0208_table() + jis0212_table(), 16, 4)
    fh.write(f"#define {prefix.upper()}_SJIS_N {SJIS_N}\n#define {prefix.upper()}_EUCKR_N {EUCKR_N}\n")
    emit_array(fh, "uint16_t", f"{prefix}_sjis", shift_jis_table(), 16, 4)
    emit_array(fh, "uint16_t", f"{prefix}_euckr", euc_kr_table(), 16, 4)
    # gb18030 / GBK: [GB_N two-byte cells][GB_RANGES breakpoint pointers][GB_RANGES code points]
    cells, ptrs, starts = gb18030_tables()
    fh.write(f"#define {prefix.upper()}_GB_N {GB_N}\n#define {prefix.upper()}_GB_RANGES {len(ptrs)}\n")
    emit_array(fh, "uint16_t", f"{prefix}_gb18030", cells + ptrs + starts, 16, 4)


if __name__ == "__main__":
    prefix = sys.argv[1] if len(sys.argv) > 1 else "sx"
    path = sys.argv[2] if len(sys.argv) > 2 else "sx_tables.inc"
    with open(path, "w") as fh:
        emit(prefix, fh)
