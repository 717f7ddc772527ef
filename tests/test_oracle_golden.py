"""Pins the ORACLE (oracle/libsxo.so) against every golden vector the reference
holds for the hot path: the three CLI golden outputs
(tests/functional/run-tests:11-41) and the in-file unit-test known answers."""
import os

import pytest

import refconfig as rc
import sxo_binding as sxo
from golden import unit_kats as K
from kat_runner import check_scan_kat, check_split_kat

G = os.path.join(os.path.dirname(__file__), "golden")


def rd(name):
    with open(os.path.join(G, name), "rb") as fh:
        return fh.read()


# tests/functional/run-tests:11-41
CLI_CASES = [
    ("expected_output1", dict(encodings=["UTF-8", "utf-16le", "utf-16be"], output_line_len="16", grep_char="63",
                              ascii_filter="All-Ctrl", unicode_block_filter="Common"), ["input1"]),
    ("expected_output2", dict(encodings=["UTF-8", "utf-16le", "utf-16be"], chars_min="10", output_line_len="32",
                              grep_char="58", ascii_filter="All-Ctrl", unicode_block_filter="Common"),
     ["input1", "input2"]),
    ("expected_output3", dict(encodings=["UTF-8", "utf-16le", "utf-16be"], output_line_len="32",
                              ascii_filter="None", unicode_block_filter="None"), ["input1", "input2"]),
]


@pytest.mark.parametrize("expected,flags,inputs", CLI_CASES, ids=[c[0] for c in CLI_CASES])
def test_cli_golden_outputs(expected, flags, inputs):
    out = sxo.run_cli(rc.missions(**flags), [rd(i) for i in inputs], radix="x")
    assert out == rd(expected)


@pytest.mark.parametrize("kat", K.SCAN_KATS, ids=[k["name"] for k in K.SCAN_KATS])
def test_scan_known_answers(kat):
    check_scan_kat(sxo.Scanner, kat)


@pytest.mark.parametrize("kat", K.SPLIT_KATS, ids=[k["src"] for k in K.SPLIT_KATS])
def test_split_str_known_answers(kat):
    check_split_kat(sxo.split_str, kat)


def test_merger_known_answer():
    k = K.MERGER_KAT
    ms = rc.missions(**k["flags"])
    per = []
    merged = []
    for m in ms:
        got = sxo.Scanner(m).scan(k["input"], file_id=0, is_last=True)
        per.append([f["s"] for f in got])
        merged += [(f["s"], f["position"], f["precision"], m["mission_id"]) for f in got]
    assert per == k["per_mission"]
    merged.sort(key=lambda t: (t[1], t[3]))  # stable: Finding::partial_cmp (finding.rs:92-109)
    assert merged == k["merged"]


def test_filter_bit_tests():
    # mission.rs:757-774
    r = sxo.split_str("A©".encode(), 1, False, False, True, rc.AF_ALL, rc.UBF_LATIN, None, 10)
    assert [x["s"] for x in r] == ["A©"]
    r = sxo.split_str("€".encode(), 1, False, False, True, rc.AF_ALL, rc.UBF_LATIN, None, 10)
    assert r == []


def test_enc_opt_parser_mirror():
    # mission.rs:776-853 (the helper in refconfig mirrors parse_enc_opt + Missions::new defaults)
    m = rc.missions(encodings=["utf-8,10,0x89AB,0xCDEF,0x2f"])[0]
    assert (m["chars_min_nb"], m["af"], m["ubf"], m["grep_char"]) == (10, 0x89AB, 0xCDEF, 0x2F)
    m = rc.missions(encodings=["ascii,10,,Latin"])[0]
    assert m["ubf"] == rc.UBF_LATIN | rc.UBF_ACCENTS and m["encoding"] == 0 and m["print_encoding_as_ascii"]
    assert rc.missions(encodings=["ascii"], unicode_block_filter="All")[0]["ubf"] == \
        rc.UBF_ALL & ~rc.UBF_INVALID & ~rc.UBF_ASIAN  # prefix match quirk: `All` -> `All-Asian`
    with pytest.raises(ValueError):
        rc.missions(encodings=["ascii,10,my-no-encoding"])
