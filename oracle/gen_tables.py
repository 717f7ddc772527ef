#!/usr/bin/env python3
"""ORACLE tables (test infrastructure): generates oracle/sxo_tables.inc from the ICU dumps in
oracle/tables/ (made by oracle/tables/dump_icu.js with node's TextDecoder), NOT from CPython's codecs —
the product's tables (stringsext_amd/csrc/gen_tables.py) come from CPython, so that the two sides of every
parity test decode with tables of different origin.  tests/test_tables.py compares the two results cell by
cell and keeps the list of raw disagreements (tests/golden/table_sources_report.txt).

encoding_rs 0.8.34 (Cargo.toml:19) is not vendored in /root/reference and there is no network: the WHATWG
index files themselves are not available.  ICU differs from them in known places, patched here:
  single byte  KOI8-U 0xAE/0xBE = U+045E/U+040E (the WHATWG index is KOI8-RU's; ICU has real KOI8-U);
               windows-874 0xDB..DE, 0xFC..FF unmapped (ICU: private use); windows-1253 0xAA unmapped
               (ICU: U+00AA); windows-1255 0xCA = U+05BA (ICU: unmapped); ISO-8859-16 is unknown to this ICU
               (row from oracle/tables/cpython_supplement.txt: single source).
  Big5         ICU maps HKSCS and ETEN's C6A1..C8FE to the private use area; the WHATWG index has the HKSCS-2008
               code points there -> those 4906 cells come from cpython_supplement.txt (single source).  The four
               pointers that decode to two code points (1133, 1135, 1164, 1166) are handled by the decoder.
  EUC-JP       jis0208: ICU's user-defined rows (private use) dropped; 8E E0..E2 (an ICU extension) dropped;
               jis0212: the IBM extension rows 0xF3/0xF4 (ICU's eucJP-ms flavour) dropped — the WHATWG index is
               JIS X 0212-1990 — and 0xA2B7 = U+FF5E as in ICU.
  Shift_JIS    ICU and CPython's cp932 agree on every two-byte cell; the single byte 0x80 is U+0080 in the WHATWG decoder
               (an error in ICU) — that is in the decoder, not in this table.
  EUC-KR       ICU has KS X 1001 only; the UHC extension (the WHATWG index is windows-949) comes from cpython_supplement.txt
               (single source).
  gb18030      ICU (GB18030-2005) has the WHATWG assignment of 0xA8BC / four-byte pointer 7457; 0xA3A0 = U+3000 is patched in (both
               sources: U+E5E5); the 18 code points GB18030-2022 moved out of the private use area are left as in both sources.
Parity of all legacy tables is UNPINNED: no reference test decodes any of them.

Order of the single-byte tables defines the encoding id (16 + index); names are Encoding::name()."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
TAB = os.path.join(HERE, "tables")

SB_NAMES = ["KOI8-R", "IBM866", "ISO-8859-2", "ISO-8859-5", "ISO-8859-15", "windows-1251", "windows-1252",
            "ISO-8859-3", "ISO-8859-4", "ISO-8859-6", "ISO-8859-7", "ISO-8859-8", "ISO-8859-8-I", "ISO-8859-10",
            "ISO-8859-13", "ISO-8859-14", "ISO-8859-16", "KOI8-U", "macintosh", "windows-874", "windows-1250",
            "windows-1253", "windows-1254", "windows-1255", "windows-1256", "windows-1257", "windows-1258",
            "x-mac-cyrillic"]
SB_PATCH = {
    "KOI8-U": {0xAE: 0x045E, 0xBE: 0x040E},
    "windows-874": {b: 0 for b in (0xDB, 0xDC, 0xDD, 0xDE, 0xFC, 0xFD, 0xFE, 0xFF)},
    "windows-1253": {0xAA: 0},
    "windows-1255": {0xCA: 0x05BA},
}
BIG5_N = 126 * 157
JIS_N = 94 * 94
SJIS_N = 11280      # pointers of the Shift_JIS decoder: (lead - offset) * 188 + trail - offset, lead up to 0xFC
EUCKR_N = 126 * 190


def is_pua(cp):
    return 0xE000 <= cp <= 0xF8FF


def single_byte():
    rows = {}
    for line in open(os.path.join(TAB, "icu_single_byte.txt")):
        p = line.split()
        if p[1] != "UNSUPPORTED":
            rows[p[0]] = [int(x, 16) for x in p[1:]]
    for line in open(os.path.join(TAB, "cpython_supplement.txt")):
        p = line.split()
        if p[0] != "big5" and p[0] not in rows:
            rows[p[0]] = [int(x, 16) for x in p[1:]]
    out = []
    for name in SB_NAMES:
        r = list(rows[name.lower()])
        for b, v in SB_PATCH.get(name, {}).items():
            r[b - 0x80] = v
        out.append(r)
    return out


def big5_pointer(lead, trail):
    return (lead - 0x81) * 157 + (trail - (0x40 if trail < 0x7F else 0x62))


def big5():
    t = [0] * BIG5_N
    for line in open(os.path.join(TAB, "icu_big5.txt")):
        k, v = line.split()
        key = int(k, 16)
        cps = [int(x, 16) for x in v.split("+")]
        if len(cps) == 1 and not is_pua(cps[0]):
            t[big5_pointer(key >> 8, key & 0xFF)] = cps[0]
    for line in open(os.path.join(TAB, "cpython_supplement.txt")):
        p = line.split()
        if p[0] != "big5":
            continue
        key = int(p[1], 16)
        cps = [int(x, 16) for x in p[2].split("+")]
        ptr = big5_pointer(key >> 8, key & 0xFF)
        if len(cps) == 1:
            t[ptr] = cps[0]
        else:
            assert ptr in (1133, 1135, 1164, 1166), hex(key)
    return t


def euc_jp():
    j208, j212 = [0] * JIS_N, [0] * JIS_N
    for line in open(os.path.join(TAB, "icu_euc_jp.txt")):
        k, v = line.split()
        cps = [int(x, 16) for x in v.split("+")]
        if len(cps) != 1 or is_pua(cps[0]):
            continue
        if len(k) == 4:
            a, b = int(k[:2], 16), int(k[2:], 16)
            if a >= 0xA1:
                j208[(a - 0xA1) * 94 + (b - 0xA1)] = cps[0]
        else:
            a, b = int(k[2:4], 16), int(k[4:], 16)
            if a not in (0xF3, 0xF4):
                j212[(a - 0xA1) * 94 + (b - 0xA1)] = cps[0]
    return j208, j212


def shift_jis():
    """index jis0208 as the Shift_JIS decoder addresses it (the user-defined pointers 8836..10715 stay 0: a rule)"""
    t = [0] * SJIS_N
    for line in open(os.path.join(TAB, "icu_shift_jis.txt")):
        k, v = line.split()
        key = int(k, 16)
        cps = [int(x, 16) for x in v.split("+")]
        lead, trail = key >> 8, key & 0xFF
        ptr = (lead - (0x81 if lead < 0xA0 else 0xC1)) * 188 + (trail - (0x40 if trail < 0x7F else 0x41))
        if len(cps) == 1 and not 8836 <= ptr <= 10715:
            t[ptr] = cps[0]
    return t


def euc_kr():
    t = [0] * EUCKR_N
    for path, pick in (("icu_euc_kr.txt", lambda p: (p[0], p[1])), ("cpython_supplement.txt", lambda p: (p[1], p[2]) if p[0] == "euc-kr" else None)):
        for line in open(os.path.join(TAB, path)):
            kv = pick(line.split())
            if not kv:
                continue
            key, cp = int(kv[0], 16), int(kv[1].split("+")[0], 16)
            if "+" in kv[1] or is_pua(cp):   # ICU's user-defined rows C9xx / FExx (private use) are not in the WHATWG index
                continue
            t[((key >> 8) - 0x81) * 190 + ((key & 0xFF) - 0x41)] = cp
    return t


GB_N = 126 * 190


def gb18030():
    """index gb18030 and its ranges from the ICU dump; 0xA3A0 = U+3000 as in the WHATWG index (ICU and CPython: U+E5E5, the
    GB18030 private-use assignment); everything else as ICU has it (0xA8BC = U+1E3F, pointer 7457 = U+E7C7: the WHATWG version)"""
    cells = [0] * GB_N
    for line in open(os.path.join(TAB, "icu_gb18030.txt")):
        k, v = line.split()
        key = int(k, 16)
        lead, trail = key >> 8, key & 0xFF
        cells[(lead - 0x81) * 190 + (trail - (0x40 if trail < 0x7F else 0x41))] = int(v, 16)
    cells[(0xA3 - 0x81) * 190 + (0xA0 - 0x41)] = 0x3000
    ptrs, starts = [], []
    for line in open(os.path.join(TAB, "icu_gb18030_ranges.txt")):
        a, b = line.split()
        assert b != "-", "a four-byte pointer below 39420 without mapping"
        ptrs.append(int(a)); starts.append(int(b, 16))
    return cells, ptrs, starts


def emit_array(fh, ctype, name, values, per_line, width):
    fh.write(f"static const {ctype} {name}[{len(values)}] = {{\n")
    for i in range(0, len(values), per_line):
        fh.write(" " + ",".join(f"0x{v:0{width}X}" for v in values[i:i + per_line]) + ",\n")
    fh.write("};\n")


def emit(path):
    sb = single_byte()
    b5 = big5()
    j208, j212 = euc_jp()
    with open(path, "w") as fh:
        fh.write("/* GENERATED by oracle/gen_tables.py from oracle/tables/ (ICU dumps) - do not edit. */\n")
        fh.write(f"#define SXO_N_SB_TABLES {len(SB_NAMES)}\n")
        fh.write(f"static const char* const sxo_sb_names[{len(SB_NAMES)}] = {{\n")
        for name in SB_NAMES:
            fh.write(f'    "{name}",\n')
        fh.write("};\n")
        fh.write(f"static const uint16_t sxo_sb_tables[{len(SB_NAMES)}][128] = {{\n")
        for name, r in zip(SB_NAMES, sb):
            fh.write(f"    /* {name} */ {{\n")
            for i in range(0, 128, 8):
                fh.write("        " + ", ".join(f"0x{v:04X}" for v in r[i:i + 8]) + ",\n")
            fh.write("    },\n")
        fh.write("};\n")
        fh.write(f"#define SXO_BIG5_N {BIG5_N}\n#define SXO_JIS_N {JIS_N}\n")
        emit_array(fh, "uint32_t", "sxo_big5", b5, 12, 5)
        emit_array(fh, "uint16_t", "sxo_jis0208", j208, 16, 4)
        emit_array(fh, "uint16_t", "sxo_jis0212", j212, 16, 4)
        fh.write(f"#define SXO_SJIS_N {SJIS_N}\n#define SXO_EUCKR_N {EUCKR_N}\n")
        emit_array(fh, "uint16_t", "sxo_sjis", shift_jis(), 16, 4)
        emit_array(fh, "uint16_t", "sxo_euckr", euc_kr(), 16, 4)
        cells, ptrs, starts = gb18030()
        fh.write(f"#define SXO_GB_N {GB_N}\n#define SXO_GB_RANGES {len(ptrs)}\n")
        emit_array(fh, "uint16_t", "sxo_gb18030", cells, 16, 4)
        emit_array(fh, "uint16_t", "sxo_gb_range_ptr", ptrs, 16, 4)
        emit_array(fh, "uint16_t", "sxo_gb_range_cp", starts, 16, 4)


if __name__ == "__main__":
    emit(sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "sxo_tables.inc"))
