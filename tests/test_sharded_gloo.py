"""Multi-process (gloo, CPU) tests of the byte-range sharding splice: N ranks, each replaying
its own byte range of one file with a halo, chained "where did you stop" exchange, gather of
the Finding buffers to rank 0 — the result must equal the oracle's single sequential scan.
Stage A is replaced by the oracle's run finder here (no GPU); the GPU run uses the kernels."""
import os
import random
import socket

import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

import refconfig as rc
import sxo_binding as sxo


def oracle_findings(mdicts, data):
    """Findings of the reference loop (4 KiB slices, k-merge per slice)."""
    scs = [sxo.Scanner(m) for m in mdicts]
    out = []
    for si, off in enumerate(range(0, len(data), 4096)):
        per = []
        for mi, sc in enumerate(scs):
            per += [(f["position"], mi, i, f) for i, f in enumerate(sc.scan(data[off:off + 4096], file_id=1))]
        per.sort(key=lambda t: (t[0], t[1], t[2]))
        out += [(f["position"], f["precision"], f["s"], f["completes"], mi, si) for _, mi, _, f in per]
    return out


def make_data(kind, seed):
    from test_host_logic import synth
    rng = random.Random(seed)
    n = 3 * (1 << 20) + 4096 * 5 + 123
    if kind == "planted":
        d = bytearray(synth(rng, n, 1 / 400))
        for w in (2, 3):
            for r in range(1, w):
                b = (n // w + 4095) // 4096 * 4096 * r
                d[b - 20:b + 30] = b"crossing-the-shard-boundary-" + b"x" * 22
                d[b + 4096 - 3:b + 4096 + 9] = "שלום עולם"[:6].encode("utf-8")[:12]
        return bytes(d)
    if kind == "c4":  # BASELINE config 4's image in small: records of the three encodings across EVERY shard edge (2 and 4 ranks)
        from test_gpu_baseline_configs import CORPUS
        d = bytearray(synth(rng, n, 1 / 3000))
        k = 0
        for w in (2, 4):
            for r in range(1, w):
                b = (n // w + 4095) // 4096 * 4096 * r
                for enc, at in (("utf-8", b - 30), ("utf-16-le", b + 4096 - 24), ("utf-16-be", b - 4096 - 10), ("utf-8", b + 128 * 3 - 7)):
                    text = (CORPUS[(k * 3 + 3) % len(CORPUS)] + " " + CORPUS[(k + 5) % len(CORPUS)])[:rng.choice([30, 64, 65, 100])]
                    rec = b"\x00\x00" + text.encode(enc) + b"\x00\x00"
                    at &= ~1
                    d[at:at + len(rec)] = rec
                    k += 1
        return bytes(d)
    if kind == "cjk":   # Big5 / EUC-JP: the token grid at a shard start needs a byte outside the lead range in the halo
        from test_dbcs import soup as dbcs_soup
        d = bytearray(dbcs_soup("big5", rng, n // 2) + dbcs_soup("euc-jp", rng, n - n // 2))
        for r in (1, 2):
            b = (n // 3 + 4095) // 4096 * 4096 * r
            d[b - 30000:b + 500] = b"\xa4" * 30500          # no token boundary for 30 KB in front of the shard start: wider halo
            d[b + 500:b + 540] = "天地玄黃宇宙洪荒日月盈昃辰宿列張".encode("big5")[:40]
        return bytes(d)
    if kind == "giant":  # one run far longer than the halo, across every boundary
        d = bytearray(synth(rng, n, 1 / 2000))
        text = bytes(rng.choice(b"abcdefghijklmnopqrstuvwxyz ") for _ in range(2 * (1 << 20)))
        d[300_000:300_000 + len(text)] = text
        return bytes(d)
    raise ValueError(kind)


def _worker(rank, world, port, kind, flags, halo, q, gather=True):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import stringsext_amd as sx
        from stringsext_amd import sharded
        from product_harness import oracle_runs_for_chunk
        data = make_data(kind, 1234)
        ms = rc.missions(**flags)
        sc = sx.Scanner(ms, device=sx.SX_HOST_ONLY)
        gathered, res = sharded.scan_sharded(
            sc, lambda lo, hi: data[lo:hi], len(data), file_id=1, halo=halo, device="cpu",
            runs_for_buffer=lambda buf, off: oracle_runs_for_chunk(ms, buf, off), gather=gather)
        key = lambda f: (f["position"], f["precision"], f["s"], f["completes"], f["mission_id"], f["slice_index"])
        if not gather:  # the findings stay distributed: rank k holds segment k; `gathered` = counts per rank
            assert gathered[rank] == len(res) and len(gathered) == world
            parts = [None] * world
            dist.gather_object(res.findings(), parts if rank == 0 else None, dst=0)
        if rank == 0:
            if gather:
                # (a rank whose result has several segments — more than 4 GiB of strings; here: SX_HOST_MERGE_SEG_BYTES — arrives as a LIST of pairs)
                parts = [sum((sharded.decode_findings(fb, ab) for fb, ab in (g if isinstance(g, list) else [g])), []) for g in gathered]
                if os.environ.get("SX_HOST_MERGE_SEG_BYTES"):
                    assert any(isinstance(g, list) for g in gathered)
                # the library's own splice (sx_shard_splice) == the Python restatement of it below
                spliced = sharded.splice(sc, gathered, len(data))
                assert [key(f) for f in spliced.findings()] == [key(f) for f in sharded.splice_order(parts, len(data))]
                if os.environ.get("SX_SPLICE_SEG_BYTES"):
                    assert sx.lib().sx_result_segments(spliced.h) > 1
                spliced.free()
            else:
                assert [sum(1 for f in p_[-o:] if o) for p_, o in zip(parts, gathered.overflow)] == gathered.overflow
            got = [key(f) for f in sharded.splice_order(parts, len(data))]
            want = oracle_findings(ms, data)
            q.put(("ok", got == want, len(got), len(want),
                   next(((a, b) for a, b in zip(got, want) if a != b), None)))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put(("err", traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


CASES = [
    (2, "planted", dict(encodings=["utf-8", "utf-16le"], chars_min="6", unicode_block_filter="African"), 1 << 16),
    (3, "planted", dict(encodings=["ascii", "utf-8"], chars_min="10", output_line_len="16"), 1 << 14),
    (2, "giant", dict(encodings=["utf-8"], chars_min="10"), 1 << 14),
    (3, "giant", dict(encodings=["ascii"], chars_min="4", output_line_len="20"), 1 << 12),
    # BASELINE config 4: the three-encoding -u African scan, byte-range sharded
    (2, "c4", dict(encodings=["utf-8", "utf-16le", "utf-16be"], chars_min="10", unicode_block_filter="African"), 1 << 16),
    (4, "c4", dict(encodings=["utf-8", "utf-16le", "utf-16be"], chars_min="10", unicode_block_filter="African"), 1 << 14),
    (4, "giant", dict(encodings=["utf-8", "utf-16le"], chars_min="10"), 1 << 13),
    (3, "cjk", dict(encodings=["big5", "euc-jp", "utf-8"], chars_min="4", unicode_block_filter="Asian"), 1 << 12),
    # round 5: BASELINE config 4's real layout — eight ranks —, and config 5's Mission set on them
    (8, "c4", dict(encodings=["utf-8", "utf-16le", "utf-16be"], chars_min="10", unicode_block_filter="African"), 1 << 14),
    (8, "giant", dict(encodings=["utf-8", "ascii"], chars_min="10"), 1 << 12),
    (8, "cjk", dict(encodings=["utf-8,,,African", "utf-16le,,,African", "utf-16be,,,African", "big5,,,Cjk", "euc-jp,,,Asian", "koi8-r,,,Cyrillic"], chars_min="10"), 1 << 13),
]


@pytest.mark.parametrize("gather", [True, False], ids=["gathered", "distributed"])
@pytest.mark.parametrize("world,kind,flags,halo", CASES, ids=[f"{c[0]}ranks-{c[1]}-{i}" for i, c in enumerate(CASES)])
def test_sharded_scan_equals_sequential(world, kind, flags, halo, gather):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, kind, flags, halo, q, gather)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
    assert res[0] == "ok", res[1]
    assert res[1], f"sharded != sequential: {res[2]} vs {res[3]} findings, first diff {res[4]}"


def test_ranks_whose_results_come_in_segments_are_gathered_and_spliced(monkeypatch):
    """BASELINE config 5 at 8 x 32 GiB yields more than 4 GiB of strings per rank: a rank's result then has several segments, each
    with its own str_off space, the gather ships them one by one and sx_shard_splice_segs puts them in order into a result of several
    segments (round 4: splice() raised).  Here the segment sizes are forced down (SX_HOST_MERGE_SEG_BYTES, SX_SPLICE_SEG_BYTES)."""
    monkeypatch.setenv("SX_HOST_MERGE_SEG_BYTES", "20000")
    monkeypatch.setenv("SX_GATHER_SEG_BYTES", "30000")
    monkeypatch.setenv("SX_SPLICE_SEG_BYTES", "50000")
    world, kind, flags, halo = 3, "planted", dict(encodings=["ascii", "utf-8"], chars_min="5"), 1 << 14
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, kind, flags, halo, q, True)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
    assert res[0] == "ok", res[1]
    assert res[1], f"sharded != sequential: {res[2]} vs {res[3]} findings, first diff {res[4]}"


# ---- a stream of several files, each sharded: the state at a file's end travels to the next file's first shard ----
def oracle_findings_files(mdicts, files):
    """the reference loop over a stream of files: one ScannerState per Mission for the whole stream (src/main.rs:150-168), the
    4 KiB grid restarting per file (src/input.rs:121-123)"""
    scs = [sxo.Scanner(m) for m in mdicts]
    out = []
    for fi, data in enumerate(files):
        rows = []
        for si, off in enumerate(range(0, len(data), 4096)):
            per = []
            for mi, sc in enumerate(scs):
                per += [(f["position"], mi, i, f) for i, f in enumerate(sc.scan(data[off:off + 4096], file_id=fi + 1))]
            per.sort(key=lambda t: (t[0], t[1], t[2]))
            rows += [(f["position"], f["precision"], f["s"], f["completes"], mi, si) for _, mi, _, f in per]
        out.append(rows)
    return out


def stream_files(seed):
    from test_host_logic import synth
    rng = random.Random(seed)
    n = 600_000
    a = bytearray(synth(rng, n + 1, 1 / 300))                      # odd length: the UTF-16 unit grid of the NEXT file starts at an odd byte
    text16 = "Բարեւ աշխարհ, שלום עולם, مرحبا بالعالم".encode("utf-16-le")
    a[-25:] = text16[:25]                                          # ... and a unit is cut by the file boundary
    b = bytearray(synth(rng, n, 1 / 300))
    b[:len(text16) - 25] = text16[25:]
    line = b"a line of text that is much longer than the sixty-four chars one output line may hold, so that it is cut and continued, "
    b[-70:] = line[:70]                                            # a string across the file boundary: its rest is a `+` line of the next file
    c = bytearray(synth(rng, n // 2 + 77, 1 / 300))
    c[:len(line) - 70] = line[70:]
    return [bytes(a), bytes(b), bytes(c)]


def _stream_worker(rank, world, port, flags, halo, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import stringsext_amd as sx
        from stringsext_amd import sharded
        from product_harness import oracle_runs_for_chunk
        files = stream_files(99)
        ms = rc.missions(**flags)
        sc = sx.Scanner(ms, device=sx.SX_HOST_ONLY)
        key = lambda f: (f["position"], f["precision"], f["s"], f["completes"], f["mission_id"], f["slice_index"])
        got, stream_off = [], 0
        for fi, data in enumerate(files):
            def runs(buf, off, data=data, stream_off=stream_off):
                return oracle_runs_for_chunk(ms, buf, stream_off + off, data[max(0, off - 4096):off])
            gathered, res = sharded.scan_sharded(sc, lambda lo, hi, data=data: data[lo:hi], len(data), file_id=fi + 1,
                                                 file_stream_off=stream_off, halo=halo, device="cpu", runs_for_buffer=runs, gather=True)
            if rank == 0:
                parts = [sharded.decode_findings(fb, ab) for fb, ab in gathered]
                got.append([key(f) for f in sharded.splice_order(parts, len(data))])
            stream_off += len(data)
        if rank == 0:
            want = oracle_findings_files(ms, files)
            q.put(("ok", got == want, [len(g) for g in got], [len(w) for w in want],
                   next(((fi, a, b) for fi, (g, w) in enumerate(zip(got, want)) for a, b in zip(g, w) if a != b), None)))
    except Exception:  # pragma: no cover
        import traceback
        q.put(("err", traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_stream_of_files_equals_sequential(world):
    """three files; the first ends in the middle of a UTF-16 unit (and has an odd length), the second ends in the middle of a
    line that the third continues: rank 0 of the next file must start from the LAST rank's state of the file before"""
    flags = dict(encodings=["utf-16le", "ascii", "utf-8"], chars_min="6", unicode_block_filter="African")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_stream_worker, args=(r, world, port, flags, 1 << 14, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
    assert res[0] == "ok", res[1]
    assert res[1], f"sharded stream != sequential: {res[2]} vs {res[3]} findings per file, first diff {res[4]}"


def _failing_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import stringsext_amd as sx
        from stringsext_amd import sharded
        data = make_data("planted", 1234)
        ms = rc.missions(encodings=["utf-8"], chars_min="6")
        sc = sx.Scanner(ms, device=sx.SX_HOST_ONLY)

        def runs(buf, off):
            if rank == 1:
                raise RuntimeError("this rank's stage A fails")   # (the library sees a failed callback)
            from product_harness import oracle_runs_for_chunk
            return oracle_runs_for_chunk(ms, buf, off)
        try:
            sharded.scan_sharded(sc, lambda lo, hi: data[lo:hi], len(data), file_id=1, halo=1 << 14, device="cpu", runs_for_buffer=runs)
            q.put((rank, "no error"))
        except Exception as e:
            q.put((rank, type(e).__name__ + ": " + str(e)[:120]))
    finally:
        dist.destroy_process_group()


def test_a_failing_rank_fails_every_rank_instead_of_hanging_them():
    """a rank that failed used to leave before the all-gather and the others waited for ever"""
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_failing_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    assert all("no error" not in v for v in got.values()), got
    assert "rank 1" in got[0] and "rank 1" in got[2], got
