#!/usr/bin/env python3
"""Cells the ORACLE's tables cannot take from ICU (oracle/tables/icu_*.txt), taken from CPython's codecs
instead — i.e. the cells that have only ONE source (the same one the product's generator uses):
  * Big5: ICU maps the whole HKSCS area (and ETEN's C6A1..C8FE) to the private use area, the WHATWG index has
    the real HKSCS code points there -> CPython `big5hkscs`; the six cells only `cp950` has (C6CF, C6D3, C6D5,
    C6D7, C6DE, C6DF).
  * EUC-KR: ICU's `euc-kr` is KS X 1001 only; the cells of the UHC extension (the WHATWG index is windows-949) -> `cp949`.
  * ISO-8859-16: unknown to ICU's TextDecoder in this image.
Run:  python3 oracle/tables/dump_cpython_supplement.py   (writes cpython_supplement.txt next to it)"""
import os

HERE = os.path.dirname(os.path.abspath(__file__))


def cps(codec, bs):
    try:
        return [ord(c) for c in bytes(bs).decode(codec)]
    except UnicodeDecodeError:
        return None


def main():
    icu = {}
    for line in open(os.path.join(HERE, "icu_big5.txt")):
        k, v = line.split()
        icu[int(k, 16)] = [int(x, 16) for x in v.split("+")]
    out = []
    for lead in range(0x81, 0xFF):
        for trail in list(range(0x40, 0x7F)) + list(range(0xA1, 0xFF)):
            key = lead << 8 | trail
            v = icu.get(key)
            if v is not None and not (0xE000 <= v[0] <= 0xF8FF):
                continue  # ICU has a real code point here
            w = cps("big5hkscs", [lead, trail]) or cps("cp950", [lead, trail])
            if w:
                out.append("big5 %04x %s" % (key, "+".join("%04x" % x for x in w)))
    # EUC-KR: ICU's table is KS X 1001 only; the WHATWG index is windows-949 (the UHC extension) -> CPython cp949
    icu_kr = set()
    for line in open(os.path.join(HERE, "icu_euc_kr.txt")):
        icu_kr.add(int(line.split()[0], 16))
    n_kr = 0
    for lead in range(0x81, 0xFF):
        for trail in range(0x41, 0xFF):
            key = lead << 8 | trail
            if key in icu_kr:
                continue
            w = cps("cp949", [lead, trail])
            if w and len(w) == 1 and not (w[0] == lead or w[0] < 0x80):
                out.append("euc-kr %04x %04x" % (key, w[0]))
                n_kr += 1
    print(n_kr, "euc-kr cells (UHC extension)")
    row = []
    for b in range(0x80, 0x100):
        w = cps("iso8859_16", [b])
        row.append("%04x" % w[0] if w else "0")
    out.append("iso-8859-16 " + " ".join(row))
    with open(os.path.join(HERE, "cpython_supplement.txt"), "w") as fh:
        fh.write("\n".join(out) + "\n")
    print(sum(1 for l in out if l.startswith("big5 ")), "big5 cells + 1 single-byte row")


if __name__ == "__main__":
    main()
