"""integration/pin_vectors.json (what integration/pin_vectors.rs feeds to the real encoding_rs at first integration) is the current
export of the hand-derived decoder vectors and of the single-source table cells."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pin_vectors_json_is_current():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import export_pin_vectors as ex
    path = os.path.join(ROOT, "integration", "pin_vectors.json")
    assert open(path, encoding="utf-8").read() == ex.text(), "regenerate with tools/export_pin_vectors.py"
    d = json.load(open(path, encoding="utf-8"))
    by = {}
    for c in d["cells"]:
        by[c[0]] = by.get(c[0], 0) + 1
    assert by["iso-8859-16"] == 128 and by["big5"] > 4000 and by["euc-kr"] > 8000, by
    assert len(d["decoder"]) >= 80 and {s["encoding"] for s in d["decoder"]} >= {"utf-8", "utf-16le", "utf-16be", "gb18030", "iso-2022-jp"}
    assert os.path.exists(os.path.join(ROOT, "integration", "pin_vectors.rs"))
