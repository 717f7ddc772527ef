"""Writes integration/pin_vectors.json: what integration/pin_vectors.rs feeds to the REAL encoding_rs 0.8.34 at first integration.
  * "decoder": every hand-derived (result, read, written) sequence of tests/golden/decoder_vectors.py;
  * "cells": the table cells this repository has from ONE source only (tests/golden/table_sources_report.txt) — Big5's HKSCS / ETEN
    cells that ICU maps to the private use area, EUC-KR's UHC extension, all of ISO-8859-16 — as (label, bytes, text) spot checks.
usage: tools/export_pin_vectors.py   (re-run when the vectors or the tables change; tests/test_pin_vectors.py checks it is current)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden.decoder_vectors import VECTORS
import test_tables as tt


def build():
    dec = [dict(name=n, encoding=e, calls=[dict(hex=h, last=bool(l), steps=[list(s) for s in steps]) for h, l, steps in calls])
           for n, e, calls in VECTORS if all(h is not None for h, _, _ in calls)]
    fin = tt._arrays("stringsext_amd/csrc/sx_tables.inc")
    tab = os.path.join(ROOT, "oracle", "tables")
    cells = []
    # ISO-8859-16 (ICU 70 does not know it)
    p = tt._load("stringsext_amd/csrc/gen_tables.py", "gt_product")
    names = [t[0] for t in p.TABLES]
    t16 = fin["sb"][names.index("ISO-8859-16")]
    for i, cp in enumerate(t16):
        if cp:
            cells.append(["iso-8859-16", "%02x" % (0x80 + i), chr(cp)])
    # Big5: cells ICU maps to the private use area and the product has (from CPython big5hkscs / cp950 alone)
    icu = {}
    for line in open(os.path.join(tab, "icu_big5.txt")):
        k, v = line.split()
        icu[int(k, 16)] = [int(x, 16) for x in v.split("+")]
    n = 126 * 157
    pb5 = [lo | (0x20000 if (fin["sx_big5"][n + (i >> 4)] >> (i & 15)) & 1 else 0) for i, lo in enumerate(fin["sx_big5"][:n])]
    for lead in range(0x81, 0xFF):
        for trail in list(range(0x40, 0x7F)) + list(range(0xA1, 0xFF)):
            ptr = (lead - 0x81) * 157 + (trail - (0x40 if trail < 0x7F else 0x62))
            a = icu.get(lead << 8 | trail)
            if a and 0xE000 <= a[0] <= 0xF8FF and pb5[ptr]:
                cells.append(["big5", "%02x%02x" % (lead, trail), chr(pb5[ptr])])
    # EUC-KR: cells only CPython cp949 has (the UHC extension)
    icu_kr = {}
    for line in open(os.path.join(tab, "icu_euc_kr.txt")):
        k, v = line.split()
        icu_kr[int(k, 16)] = int(v, 16)
    for lead in range(0x81, 0xFF):
        for trail in range(0x41, 0xFF):
            a = icu_kr.get(lead << 8 | trail, 0)
            a = 0 if 0xE000 <= a <= 0xF8FF else a
            cp = fin["sx_euckr"][(lead - 0x81) * 190 + trail - 0x41]
            if cp and not a:
                cells.append(["euc-kr", "%02x%02x" % (lead, trail), chr(cp)])
    return dict(crate="encoding_rs 0.8.34 (Cargo.lock:147-150 of getreu/stringsext v2.3.5)", decoder=dec, cells=cells)


def text():
    return json.dumps(build(), ensure_ascii=False, indent=0, separators=(",", ":")) + "\n"


if __name__ == "__main__":
    out = os.path.join(ROOT, "integration", "pin_vectors.json")
    open(out, "w", encoding="utf-8").write(text())
    d = build()
    print(f"{out}: {len(d['decoder'])} decoder sequences, {len(d['cells'])} single-source cells")
