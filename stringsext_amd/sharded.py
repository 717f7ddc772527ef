"""Byte-range sharding of one input file over the GPUs of a node: one process per GPU,
`torch.distributed` for the two tiny exchanges the path really has —

  1. where each rank's replay started and stopped (two u64 per mission + a count, one
     all_gather; a rank whose predecessor ran past the point it started from repeats its replay
     from there — rare — and the table is exchanged again), and
  2. optionally the gather of the Finding buffers to rank 0 (RCCL over xGMI with backend
     "nccl"); by default the findings stay distributed, rank k holding segment k.

There is no collective on the data path: every rank scans its own byte range (plus a halo)
with the same kernels as the single-GPU path.  The reference has nothing comparable (one
thread per Mission, src/main.rs:97-151); the splice reproduces what its single sequential
stream would have printed.
"""
import ctypes
import os
import struct
import sys
import time

import torch
import torch.distributed as dist

from . import Finding, PRECISION, SxError, SX_E_HALO


class _NoResult:
    def free(self):
        pass

HALO_DEFAULT = 1 << 20


def shard_bounds(file_len, world, rank):
    """[own_lo, own_hi): contiguous, on the 4096-byte slice grid (src/input.rs:22)."""
    per = (file_len // world + 4095) // 4096 * 4096
    lo = min(file_len, rank * per)
    hi = file_len if rank == world - 1 else min(file_len, (rank + 1) * per)
    return lo, hi


def _all_gather_u64(values, device):
    t = torch.tensor(values, dtype=torch.int64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [[int(x) for x in o.tolist()] for o in out]


def _payload(res):
    """The rank's findings as ONE (findings bytes, arena bytes) pair without going through Python bytes: numpy views
    of the result's segments (pinned host memory the device wrote), str_off rebased where there are several."""
    import numpy as np
    fdt = np.dtype({"names": ["position", "str_off", "str_len", "slice_index"], "formats": ["<u8", "<u4", "<u4", "<u4"],
                    "offsets": [0, 8, 12, 24], "itemsize": ctypes.sizeof(Finding)})
    assert fdt.itemsize == ctypes.sizeof(Finding)
    fs, ars, base = [], [], 0
    for fp, n, ap, alen in res.segment_pointers():
        if n:
            f = np.ctypeslib.as_array(ctypes.cast(fp, ctypes.POINTER(ctypes.c_uint8)), shape=(n * fdt.itemsize,)).view(fdt)
            if base:
                f = f.copy()
                f["str_off"] += base
            fs.append(f)
        if alen:
            ars.append(np.ctypeslib.as_array(ap, shape=(alen,)))
        base += alen
    if base > 0xFFFFFFFF:
        raise ValueError("more than 4 GiB of strings in one rank's result: gather segment by segment")
    f = fs[0] if len(fs) == 1 else (np.concatenate(fs) if fs else np.zeros(0, fdt))
    a = ars[0] if len(ars) == 1 else (np.concatenate(ars) if ars else np.zeros(0, np.uint8))
    return f.view(np.uint8), a


def scan_sharded(scanner, get_buffer, file_len, file_id=1, file_stream_off=0, halo=HALO_DEFAULT, device="cpu",
                 runs_for_buffer=None, gather=True, timings=None):
    """Scan one file of `file_len` bytes sharded over the process group.

    get_buffer(lo, hi) -> bytes | ctypes.c_void_p : the file bytes [lo, hi) on this rank.
    runs_for_buffer(buf_bytes, buf_off) -> runs per mission (tests on CPU: stage B only).
    gather=True: rank 0 gets the list of (findings_bytes, arena_bytes) per rank, in rank order (a
    gather over the process group), the other ranks None.  gather=False: the findings stay where
    they are — rank k's Result is segment k of the file's findings, in order — and every rank
    gets the per-rank finding counts (ShardCounts).  The rank's own Result is the second value.
    Segment k may end with a few findings that lie behind rank k's range end (a region across the
    boundary; ShardCounts.overflow[k] of them); splice_order() merges them into segment k+1's head.
    """
    world, rank = dist.get_world_size(), dist.get_rank()
    own_lo, own_hi = shard_bounds(file_len, world, rank)
    nm = scanner.n

    def attempt(start_at, h, reuse):
        buf_lo = max(0, own_lo - h) // 4096 * 4096
        buf_hi = min(file_len, own_hi + h)
        buf = get_buffer(buf_lo, buf_hi)
        kw = {}
        if runs_for_buffer is not None:
            kw["runs_per_mission"] = runs_for_buffer(buf, buf_lo)
        if isinstance(buf, ctypes.c_void_p):
            kw["buf_len"] = buf_hi - buf_lo
        try:
            res, ends = scanner.scan_shard(buf, buf_lo, own_lo, own_hi, start_at=start_at, file_stream_off=file_stream_off,
                                           file_id=file_id, reuse_runs=reuse, **kw)
        except SxError as e:
            if e.code != SX_E_HALO or buf_lo == 0:
                raise
            return _NoResult(), [own_hi] * nm, True, buf_hi   # the halo in front is too short (Big5 / EUC-JP): look further
        truncated = any(e >= buf_hi for e in ends) and buf_hi < file_len
        return res, ends, truncated, buf_hi

    # first attempt: everybody assumes the previous rank stops at the shard boundary
    t_begin = time.perf_counter()
    h = halo
    start = [own_lo] * nm
    res, ends, truncated, _ = attempt(None, h, False)
    while truncated:  # a run crosses the whole halo: look further
        h *= 8
        res.free()
        res, ends, truncated, _ = attempt(None, h, False)

    # Where did everybody stop?  One all_gather of (start used, end reached) per mission + the
    # finding count; rank k must repeat its replay if rank k-1 ran past the point rank k started
    # from (rare: a region crossing the shard boundary).  Every rank evaluates the same table,
    # so all agree on who repeats; a repeat can move that rank's own end, hence the loop.
    t_scan = time.perf_counter()
    while True:
        table = _all_gather_u64(start + ends + [len(res)], device)
        redo = [False] * world
        for k in range(1, world):
            prev_end, my_start = table[k - 1][nm:2 * nm], table[k][:nm]
            k_lo = shard_bounds(file_len, world, k)[0]
            redo[k] = any(max(k_lo, p) > s0 for p, s0 in zip(prev_end, my_start))
        if not any(redo):
            break
        if redo[rank]:
            prev_end = table[rank - 1][nm:2 * nm]
            start = [max(own_lo, p, s0) for p, s0 in zip(prev_end, start)]
            res.free()
            res, ends, truncated, _ = attempt(start, h, True)
            while truncated:
                h *= 8
                res.free()
                res, ends, truncated, _ = attempt(start, h, False)
            ends = [max(e, p) for e, p in zip(ends, prev_end)]
    # A region that crosses the shard boundary is finished by the rank it began on, so the tail of rank
    # k's findings can lie in slices that belong to rank k+1 and interleaves there with other Missions'
    # findings of rank k+1 (the reference prints slice by slice, src/main.rs:153-168).  How many such
    # findings each rank holds goes along with the counts; splice_order() puts them in place.
    over = _overflow(res, own_hi // 4096) if rank + 1 < world else 0
    counts = ShardCounts(row[2 * nm] for row in table)
    counts.overflow = [row[0] for row in _all_gather_u64([over], device)]
    if os.environ.get("SX_TIMING") and rank == 0:
        print(f"[sx] sharded: scan {1e3 * (t_scan - t_begin):.2f} ms, exchange {1e3 * (time.perf_counter() - t_scan):.2f} ms",
              file=sys.stderr)

    t_exchanged = time.perf_counter()
    if timings is not None:
        timings["scan_ms"] = 1e3 * (t_scan - t_begin)
        timings["exchange_ms"] = 1e3 * (t_exchanged - t_scan)
        timings["gather_ms"] = 0.0
    if not gather:
        return counts, res
    # The gather of the Finding buffers to rank 0 (backend "nccl" = RCCL over xGMI): sizes first, then one
    # gather of buffers padded to the largest.  The payload goes from the result's pinned memory straight into
    # the exchange tensor (no Python-level copy).
    fb, ab = _payload(res)
    sizes = _all_gather_u64([len(fb), len(ab)], device)
    mx = max(s[0] + s[1] for s in sizes)
    mine = torch.empty(max(mx, 1), dtype=torch.uint8, device=device)
    if len(fb):
        mine[:len(fb)].copy_(torch.from_numpy(fb), non_blocking=True)
    if len(ab):
        mine[len(fb):len(fb) + len(ab)].copy_(torch.from_numpy(ab), non_blocking=True)
    out = None
    if rank == 0:
        bufs = [torch.empty(max(mx, 1), dtype=torch.uint8, device=device) for _ in range(world)]
        dist.gather(mine, bufs, dst=0)
        out = []
        for b, (nf, na) in zip(bufs, sizes):
            raw = b[:nf + na].cpu().numpy()
            out.append((raw[:nf].tobytes(), raw[nf:nf + na].tobytes()))
    else:
        dist.gather(mine, None, dst=0)
    if timings is not None:
        timings["gather_ms"] = 1e3 * (time.perf_counter() - t_exchanged)
    return out, res


class ShardCounts(list):
    """Findings per rank; .overflow[k] = how many of rank k's last findings lie behind its range end."""
    overflow = None


def _overflow(res, boundary_slice):
    n = 0
    for v, cnt in reversed(res.finding_arrays()):
        i = cnt
        while i > 0 and v[i - 1].slice_index >= boundary_slice:
            i -= 1
        n += cnt - i
        if i > 0:
            break
    return n


def splice_order(parts, file_len, key=lambda f: (f["slice_index"], f["position"])):
    """parts[k] = rank k's findings (decoded, in that rank's order) -> one list in the reference's order:
    rank k's findings behind its range end are merged into the head of rank k+1's (same rule as the
    library's Mission merge: slice, position, then Mission order; both sides are already sorted)."""
    world = len(parts)
    out = []
    carry = []  # findings of earlier ranks that lie in slices of the current one
    for k, part in enumerate(parts):
        b = shard_bounds(file_len, world, k)[1] // 4096
        cut = len(part)
        while k + 1 < world and cut > 0 and part[cut - 1]["slice_index"] >= b:
            cut -= 1
        own, nxt = part[:cut], part[cut:]
        if carry:  # two sorted lists; ties: lower Mission first, and within a Mission the earlier rank
            merged, i, j = [], 0, 0
            while i < len(carry) and j < len(own):
                a, c = carry[i], own[j]
                if (key(a), a["mission_id"]) <= (key(c), c["mission_id"]):
                    merged.append(a); i += 1
                else:
                    merged.append(c); j += 1
            own = merged + carry[i:] + own[j:]
            # what was carried may itself lie behind this rank's end (a region across a whole shard)
            cut2 = len(own)
            while k + 1 < world and cut2 > 0 and own[cut2 - 1]["slice_index"] >= b:
                cut2 -= 1
            nxt = sorted(own[cut2:] + nxt, key=lambda f: (key(f), f["mission_id"]))
            own = own[:cut2]
        out += own
        carry = nxt
    return out + carry


def decode_findings(findings_bytes, arena_bytes):
    n = len(findings_bytes) // ctypes.sizeof(Finding)
    arr = (Finding * n).from_buffer_copy(findings_bytes)
    return [dict(position=f.position, precision=PRECISION[f.precision],
                 s=arena_bytes[f.str_off:f.str_off + f.str_len].decode("utf-8"), completes=bool(f.completes_previous),
                 mission_id=f.mission_id, file_id=f.input_file_id, slice_index=f.slice_index) for f in arr]
