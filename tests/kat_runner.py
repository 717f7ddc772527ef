"""Shared assertions: run the reference's known-answer scenarios against any
backend exposing the reference's test-visible surface (a `Scanner` with
.scan/.consumed_bytes/.maybe_cut/.leftover/.first_byte_position/.arena and a
`split_str` function)."""


def check_scan_kat(make_scanner, kat):
    sc = make_scanner(kat["mission"])
    for i, call in enumerate(kat["calls"]):
        got = sc.scan(call["input"], file_id=0, is_last=call["is_last"])
        slim = [dict(position=f["position"], precision=f["precision"], s=f["s"]) for f in got]
        where = f"{kat['name']} call {i} ({kat['src']})"
        if "findings" in call:
            assert slim == call["findings"], where
        if "findings_prefix" in call:
            k = len(call["findings_prefix"])
            assert slim[:k] == call["findings_prefix"], where
        if "n_findings_not" in call:
            assert len(slim) != call["n_findings_not"], where
        if "first_byte_position" in call:
            assert sc.first_byte_position == call["first_byte_position"], where
        if "consumed" in call:
            assert sc.consumed_bytes == call["consumed"], where
        if "maybe_cut" in call:
            assert sc.maybe_cut == call["maybe_cut"], where
        if "leftover" in call:
            assert sc.leftover == call["leftover"], where
        if "arena_prefix" in call:
            ap = call["arena_prefix"]
            arena = (sc.arena + b"\0" * len(ap))[:len(ap)]
            assert arena == ap, where


def check_split_kat(split_str, kat):
    inp = kat["inp"].encode("utf-8")
    n, same, last_cut, inv, grep, q = kat["args"]
    q = len(inp) if q is None else q
    got = split_str(inp, n, same, last_cut, inv, kat["filt"]["af"], kat["filt"]["ubf"], grep, q)
    assert [g["s"] for g in got] == [e["s"] for e in kat["out"]], kat["src"]
    for g, e in zip(got, kat["out"]):
        for key, val in e.items():
            assert g[key] == val, (kat["src"], e["s"], key)
