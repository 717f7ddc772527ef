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

from . import Finding, PRECISION, Result, Run, SX_OK, SxError, lib

HALO_DEFAULT = 1 << 20


def shard_bounds(file_len, world, rank):
    """[own_lo, own_hi): contiguous, on the 4096-byte slice grid (src/input.rs:22) — sx_shard_bounds."""
    L = lib()
    L.sx_shard_bounds.argtypes = [ctypes.c_uint64, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
    L.sx_shard_bounds.restype = None
    lo, hi = ctypes.c_uint64(), ctypes.c_uint64()
    L.sx_shard_bounds(file_len, world, rank, ctypes.byref(lo), ctypes.byref(hi))
    return lo.value, hi.value


def splice(scanner, gathered, file_len):
    """gathered = [(findings_bytes, arena_bytes)] per rank (what scan_sharded(gather=True) returns on rank 0) -> one
    Result in the reference's print order (sx_shard_splice)."""
    L = lib()
    world = len(gathered)
    # (round 5) a rank with more than 4 GiB of strings arrives segment by segment, each with its own str_off space: a LIST of
    # (findings, arena) pairs for that rank — sx_shard_splice_segs takes every rank's segments in a row and yields as many result
    # segments as the strings need (round 4 refused such a rank)
    segs, per_rank = [], []
    for g in gathered:
        mine = g if isinstance(g, list) else [g]
        per_rank.append(len(mine))
        segs += [(bytes(fb), bytes(ab)) for fb, ab in mine]   # (numpy views of the gather buffer, or bytes)
    n = len(segs)
    keep = [(ctypes.create_string_buffer(fb, len(fb)), ctypes.create_string_buffer(ab, len(ab))) for fb, ab in segs]
    fptr = (ctypes.POINTER(Finding) * n)(*[ctypes.cast(f, ctypes.POINTER(Finding)) for f, _ in keep])
    aptr = (ctypes.POINTER(ctypes.c_uint8) * n)(*[ctypes.cast(a, ctypes.POINTER(ctypes.c_uint8)) for _, a in keep])
    nf = (ctypes.c_uint64 * n)(*[len(fb) // ctypes.sizeof(Finding) for fb, _ in segs])
    na = (ctypes.c_uint64 * n)(*[len(ab) for _, ab in segs])
    cnt = (ctypes.c_uint32 * world)(*per_rank)
    L.sx_shard_splice_segs.argtypes = [ctypes.POINTER(ctypes.POINTER(Finding)), ctypes.POINTER(ctypes.c_uint64),
                                       ctypes.POINTER(ctypes.POINTER(ctypes.c_uint8)), ctypes.POINTER(ctypes.c_uint64),
                                       ctypes.POINTER(ctypes.c_uint32), ctypes.c_int, ctypes.c_uint64, ctypes.POINTER(ctypes.c_void_p)]
    r = ctypes.c_void_p()
    scanner._chk(L.sx_shard_splice_segs(fptr, nf, aptr, na, cnt, world, file_len, ctypes.byref(r)))
    return Result(scanner, r)


_HOST = {}
_XCHG = {}   # (row bytes, world, device) -> the exchange's tensors


def _host_buffer(n, device):
    """a host tensor of at least n bytes, pinned where there is a GPU, reused from call to call"""
    pin = torch.device(device).type == "cuda"
    t = _HOST.get(pin)
    if t is None or t.numel() < n:
        t = torch.empty(n + n // 4 + 4096, dtype=torch.uint8, pin_memory=pin)
        _HOST[pin] = t
    return t


_DEV = {}


def _device_buffer(kind, n, device):
    """a device tensor of at least n bytes for the gather (kind: "send" / "recv"), kept and grown from call to call"""
    key = (kind, str(device))
    t = _DEV.get(key)
    if t is None or t.numel() < n:
        _DEV[key] = None
        t = _DEV[key] = torch.empty(n + n // 4 + 4096, dtype=torch.uint8, device=device)
    return t


def _all_gather_u64(values, device):
    t = torch.tensor(values, dtype=torch.int64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [[int(x) for x in o.tolist()] for o in out]


MAX_SEGMENTS = 16   # of one rank's result that the gather can ship separately (a segment holds up to 4 GiB of strings)


def _payload(res):
    """The rank's findings as (findings bytes, arena bytes) pairs without going through Python bytes: numpy views of the
    result's segments (pinned host memory the device wrote).  ONE pair (str_off rebased where there are several segments)
    as long as the strings fit 32-bit offsets, else one pair per segment, each with its own str_off space."""
    import numpy as np
    fdt = np.dtype({"names": ["position", "str_off", "str_len", "slice_index"], "formats": ["<u8", "<u4", "<u4", "<u4"],
                    "offsets": [0, 8, 12, 24], "itemsize": ctypes.sizeof(Finding)})
    assert fdt.itemsize == ctypes.sizeof(Finding)
    segs = []
    for fp, n, ap, alen in res.segment_pointers():
        f = (np.ctypeslib.as_array(ctypes.cast(fp, ctypes.POINTER(ctypes.c_uint8)), shape=(n * fdt.itemsize,)).view(fdt)
             if n else np.zeros(0, fdt))
        a = np.ctypeslib.as_array(ap, shape=(alen,)) if alen else np.zeros(0, np.uint8)
        if n or alen:
            segs.append((f, a))
    limit = int(os.environ.get("SX_GATHER_SEG_BYTES", 0xFFFFFFFF))   # (tests: ranks that ship several segments without gigabytes of strings)
    if sum(len(a) for _, a in segs) <= limit:
        # (copies and the concatenation go through the records' BYTES: the dtype above names four of a record's fields, and numpy need
        # not carry the bytes between named fields — precision, flags, Mission — through a structured copy; round 5: it did not)
        fs, ars, base = [], [], 0
        for f, a in segs:
            fb = f.view(np.uint8)
            if base and len(f):
                fb = fb.copy()
                fb.view(fdt)["str_off"] += base
            fs.append(fb); ars.append(a); base += len(a)
        f = fs[0] if len(fs) == 1 else (np.concatenate(fs) if fs else np.zeros(0, np.uint8))
        a = ars[0] if len(ars) == 1 else (np.concatenate(ars) if ars else np.zeros(0, np.uint8))
        return [(f, a)]
    if len(segs) > MAX_SEGMENTS:
        raise ValueError(f"{len(segs)} result segments on one rank: more than the gather ships ({MAX_SEGMENTS})")
    return [(f.view(np.uint8), a) for f, a in segs]


ALLGATHER_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p)
BUFFER_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.POINTER(ctypes.c_void_p),
                             ctypes.POINTER(ctypes.c_int))
RUNS_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64,
                           ctypes.POINTER(ctypes.POINTER(ctypes.POINTER(Run))), ctypes.POINTER(ctypes.POINTER(ctypes.c_uint64)))


def scan_sharded(scanner, get_buffer, file_len, file_id=1, file_stream_off=0, halo=HALO_DEFAULT, device="cpu",
                 runs_for_buffer=None, gather=True, timings=None):
    """Scan one file of `file_len` bytes sharded over the process group.  The protocol (own range + halo, the "where
    did everybody stop" exchange, repeats, wider halos) is the library's — sx_scan_sharded behind the C-ABI; this
    function only supplies the transport (torch.distributed all_gather; backend "nccl" = RCCL over xGMI) and the buffers.

    get_buffer(lo, hi) -> bytes | ctypes.c_void_p : the file bytes [lo, hi) on this rank (c_void_p: in HBM).
    runs_for_buffer(buf_bytes, buf_off) -> runs per mission (tests on CPU: stage B only).
    gather=True: rank 0 gets the list of (findings_bytes, arena_bytes) per rank, in rank order (sizes, then transfers of
    exactly those sizes), the other ranks None.  A rank whose strings exceed 4 GiB arrives as a LIST of such pairs, one per
    result segment (str_off is per segment).  gather=False: the findings stay where they are — rank k's Result is
    segment k of the file's findings, in order — and every rank gets the per-rank finding counts (ShardCounts).
    The rank's own Result is the second value.  Segment k may end with a few findings that lie behind rank k's range
    end (a region across the boundary; ShardCounts.overflow[k] of them); splice() / splice_order() merge them into
    segment k+1's head.
    """
    world, rank = dist.get_world_size(), dist.get_rank()
    L = lib()
    keep = {}
    errors = []

    on_gpu = torch.device(device).type == "cuda"

    def _allgather(user, send, nbytes, recv):
        # the exchanged rows are a few hundred bytes: one pinned host tensor per direction, kept between calls (and one device
        # tensor each for backend "nccl"), filled and read with memmove — no Python-level copies of the payload
        t_in = time.perf_counter()
        try:
            key = (nbytes, world, str(device))
            t = _XCHG.get(key)
            if t is None:
                hs = torch.empty(nbytes, dtype=torch.uint8, pin_memory=on_gpu)
                hr = torch.empty(nbytes * world, dtype=torch.uint8, pin_memory=on_gpu)
                t = _XCHG[key] = (hs, hr, hs.to(device) if on_gpu else hs, hr.to(device) if on_gpu else hr)
            hs, hr, ds, dr = t
            ctypes.memmove(hs.data_ptr(), send, nbytes)
            if on_gpu:
                ds.copy_(hs, non_blocking=True)
            dist.all_gather_into_tensor(dr, ds)
            if on_gpu:
                hr.copy_(dr, non_blocking=True)
                torch.cuda.current_stream(device).synchronize()
            ctypes.memmove(recv, hr.data_ptr(), nbytes * world)
            keep["exchange_s"] = keep.get("exchange_s", 0.0) + (time.perf_counter() - t_in)
            return 0
        except Exception as e:  # pragma: no cover
            errors.append(e)
            return 1

    def _buffer(user, lo, hi, ptr, is_device):
        try:
            buf = get_buffer(lo, hi)
            if isinstance(buf, ctypes.c_void_p):
                ptr[0] = buf.value
                is_device[0] = 1
            else:
                keep["buf"] = ctypes.create_string_buffer(bytes(buf), hi - lo)
                ptr[0] = ctypes.cast(keep["buf"], ctypes.c_void_p).value
                is_device[0] = 0
            return 0
        except Exception as e:  # pragma: no cover
            errors.append(e)
            return 1

    def _runs(user, bytes_ptr, buf_off, buf_len, runs_out, n_out):
        try:
            per = runs_for_buffer(ctypes.string_at(bytes_ptr, buf_len), buf_off)
            arrs = [(Run * max(1, len(rs)))(*[Run(*t) for t in rs]) for rs in per]
            keep["runs"] = (arrs, (ctypes.POINTER(Run) * len(per))(*[ctypes.cast(a, ctypes.POINTER(Run)) for a in arrs]),
                            (ctypes.c_uint64 * len(per))(*[len(rs) for rs in per]))
            runs_out[0] = ctypes.cast(keep["runs"][1], ctypes.POINTER(ctypes.POINTER(Run)))
            n_out[0] = ctypes.cast(keep["runs"][2], ctypes.POINTER(ctypes.c_uint64))
            return 0
        except Exception as e:  # pragma: no cover
            errors.append(e)
            return 1

    cb_gather, cb_buffer = ALLGATHER_FN(_allgather), BUFFER_FN(_buffer)
    cb_runs = RUNS_FN(_runs) if runs_for_buffer is not None else ctypes.cast(None, RUNS_FN)
    L.sx_scan_sharded.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int,
                                  ctypes.c_uint64, BUFFER_FN, ctypes.c_void_p, RUNS_FN, ctypes.c_void_p, ALLGATHER_FN, ctypes.c_void_p,
                                  ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
    t_begin = time.perf_counter()
    r = ctypes.c_void_p()
    cnt, over = (ctypes.c_uint64 * world)(), (ctypes.c_uint64 * world)()
    rc = L.sx_scan_sharded(scanner.h, rank, world, file_len, file_stream_off, file_id, halo, cb_buffer, None, cb_runs, None,
                           cb_gather, None, ctypes.byref(r), cnt, over)
    if errors:
        raise errors[0]
    scanner._chk(rc)
    res = Result(scanner, r)
    counts = ShardCounts(cnt)
    counts.overflow = list(over)
    t_scan = time.perf_counter()   # (the exchange is inside the library call: two small all-gathers)

    t_exchanged = time.perf_counter()
    if timings is not None:
        timings["scan_ms"] = 1e3 * (t_scan - t_begin)     # incl. the small all-gathers inside sx_scan_sharded ...
        timings["exchange_ms"] = 1e3 * keep.get("exchange_s", 0.0)   # ... which take this long (waiting for the slowest rank included)
        timings["gather_ms"] = 0.0
    if not gather:
        return counts, res
    # The gather of the Finding buffers to rank 0 (backend "nccl" = RCCL over xGMI): every rank's segment sizes first (one
    # all_gather of a small table), then point-to-point transfers of exactly those sizes — nothing is padded to the largest
    # payload, rank 0 holds one staging buffer on the device (the largest single segment), and a rank with more than 4 GiB of
    # strings ships its result segment by segment.  The payload goes from the result's pinned memory straight into the
    # exchange tensor (no Python-level copy).
    pairs = _payload(res)
    row = [len(pairs)] + [x for fb, ab in pairs for x in (len(fb), len(ab))] + [0] * (2 * (MAX_SEGMENTS - len(pairs)))
    sizes = _all_gather_u64(row, device)
    out = None
    if rank == 0:
        # every receive is posted at once (round 3: world - 1 receives one after the other, each through a staging buffer): the
        # senders' transfers overlap; on the GPU they land in ONE device buffer laid out like the host buffer's remote part, which a
        # single copy brings to the host
        own_total = sum(sizes[0][1:1 + 2 * sizes[0][0]])
        remote_total = sum(sum(r[1:1 + 2 * r[0]]) for r in sizes[1:])
        host = _host_buffer(max(own_total + remote_total, 1), device)
        stage = _device_buffer("recv", remote_total, device) if on_gpu and remote_total else None
        off, where, reqs = 0, [], []
        for k, r in enumerate(sizes):
            segs_k = []
            for j in range(r[0]):
                nf, na = r[1 + 2 * j], r[2 + 2 * j]
                if k == 0:
                    dst = host[off:off + nf + na]
                    fb, ab = pairs[j]
                    if nf: dst[:nf].copy_(torch.from_numpy(fb))
                    if na: dst[nf:].copy_(torch.from_numpy(ab))
                elif nf + na:
                    dst = stage[off - own_total:off - own_total + nf + na] if on_gpu else host[off:off + nf + na]
                    reqs.append(dist.P2POp(dist.irecv, dst, k))
                segs_k.append((off, nf, na))
                off += nf + na
            where.append(segs_k)
        # (one group: with RCCL the receives from different ranks then run next to each other instead of one kernel after the other)
        for q in (dist.batch_isend_irecv(reqs) if reqs else []):
            q.wait()
        if on_gpu and remote_total:
            host[own_total:own_total + remote_total].copy_(stage[:remote_total], non_blocking=True)
            torch.cuda.current_stream(device).synchronize()
        raw = host.numpy()   # the views are valid until the next gather
        out = []
        for segs_k in where:
            views = [(raw[o:o + nf], raw[o + nf:o + nf + na]) for o, nf, na in segs_k]
            out.append(views[0] if len(views) == 1 else (views if views else (raw[0:0], raw[0:0])))
    else:
        # The same API as the receiving side (round 5): rank 0 posts its receives through batch_isend_irecv, and with
        # ProcessGroupNCCL batched point-to-point runs on the group's communicator while a plain dist.send creates a two-rank communicator
        # that BOTH peers must initialise — rank 0 never would, and the gather would hang with two or more real GPUs.  All segments of
        # this rank go in one batch, each from its own slice of the send buffer (a segment's tensor must stay untouched until the batch is done).
        total = sum(len(fb) + len(ab) for fb, ab in pairs)
        ops, off = [], 0
        buf = _device_buffer("send", total, device) if on_gpu and total else None   # one send buffer, kept from call to call
        for fb, ab in pairs:
            n = len(fb) + len(ab)
            if not n:
                continue
            if on_gpu:
                mine = buf[off:off + n]
                if len(fb): mine[:len(fb)].copy_(torch.from_numpy(fb), non_blocking=True)
                if len(ab): mine[len(fb):].copy_(torch.from_numpy(ab), non_blocking=True)
            else:
                import numpy as np
                mine = torch.from_numpy(np.concatenate([fb, ab]) if len(ab) and len(fb) else (fb if len(fb) else ab))
            off += n
            ops.append(dist.P2POp(dist.isend, mine, 0))
        for q in (dist.batch_isend_irecv(ops) if ops else []):
            q.wait()
    if timings is not None:
        timings["gather_ms"] = 1e3 * (time.perf_counter() - t_exchanged)
    return out, res


# ---- the transport inside the library (round 6: csrc/sx_transport.cpp — RCCL through dlopen; the C-ABI a Rust host binds) ----
_LIB_TRANSPORT = {}


def library_transport(device_index, id_exchange=None):
    """The library's own RCCL transport for this rank (sx_transport_rccl_create), made once per process and device.  Rank 0's
    unique id travels through `id_exchange(id_bytes_or_None) -> id_bytes` (default: a broadcast over the torch.distributed group
    the launcher set up — any backend: it only ships 128 bytes once)."""
    world, rank = dist.get_world_size(), dist.get_rank()
    key = (device_index, world, rank)
    if key in _LIB_TRANSPORT:
        return _LIB_TRANSPORT[key]
    L = lib()
    L.sx_transport_rccl_id.argtypes = [ctypes.c_char_p]
    L.sx_transport_rccl_create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_char_p]
    L.sx_transport_last_error.restype = ctypes.c_char_p
    L.sx_transport_last_error.argtypes = [ctypes.c_void_p]
    ident = None
    if rank == 0:
        buf = ctypes.create_string_buffer(128)
        rc = L.sx_transport_rccl_id(buf)
        if rc != SX_OK:
            raise SxError(rc, L.sx_transport_last_error(None).decode())
        ident = buf.raw
    if id_exchange is not None:
        ident = id_exchange(ident)
    elif world > 1:
        box = [ident]
        dist.broadcast_object_list(box, src=0)
        ident = box[0]
    t = ctypes.c_void_p()
    rc = L.sx_transport_rccl_create(ctypes.byref(t), device_index, rank, world, ident)
    if rc != SX_OK:
        raise SxError(rc, L.sx_transport_last_error(None).decode())
    _LIB_TRANSPORT[key] = t
    return t


def scan_sharded_library(scanner, get_buffer, file_len, device_index, file_id=1, halo=HALO_DEFAULT, file_stream_off=0, gather=True,
                         timings=None, id_exchange=None):
    """scan_sharded with the LIBRARY's transport: sx_scan_sharded gets sx_transport_allgather as its callback (no Python in the
    exchange), sx_transport_gather brings every rank's segments to rank 0 and splices them there.  Returns (the file's findings as one
    Result on rank 0 — None elsewhere or without gather —, this rank's own Result, ShardCounts)."""
    world, rank = dist.get_world_size(), dist.get_rank()
    L = lib()
    tr = library_transport(device_index, id_exchange)
    keep, errors = {}, []

    def _buffer(user, lo, hi, ptr, is_device):
        try:
            buf = get_buffer(lo, hi)
            if isinstance(buf, ctypes.c_void_p):
                ptr[0] = buf.value
                is_device[0] = 1
            else:
                keep["buf"] = ctypes.create_string_buffer(bytes(buf), hi - lo)
                ptr[0] = ctypes.cast(keep["buf"], ctypes.c_void_p).value
                is_device[0] = 0
            return 0
        except Exception as e:  # pragma: no cover
            errors.append(e)
            return 1

    cb_buffer = BUFFER_FN(_buffer)
    cb_gather = ctypes.cast(L.sx_transport_allgather, ALLGATHER_FN)
    L.sx_scan_sharded.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int,
                                  ctypes.c_uint64, BUFFER_FN, ctypes.c_void_p, RUNS_FN, ctypes.c_void_p, ALLGATHER_FN, ctypes.c_void_p,
                                  ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
    t0 = time.perf_counter()
    r = ctypes.c_void_p()
    cnt, over = (ctypes.c_uint64 * world)(), (ctypes.c_uint64 * world)()
    rc = L.sx_scan_sharded(scanner.h, rank, world, file_len, file_stream_off, file_id, halo, cb_buffer, None, ctypes.cast(None, RUNS_FN), None,
                           cb_gather, tr, ctypes.byref(r), cnt, over)
    if errors:
        raise errors[0]
    scanner._chk(rc)
    res = Result(scanner, r)
    counts = ShardCounts(cnt)
    counts.overflow = list(over)
    t1 = time.perf_counter()
    whole = None
    if gather:
        L.sx_transport_gather.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64, ctypes.POINTER(ctypes.c_void_p)]
        out = ctypes.c_void_p()
        rc = L.sx_transport_gather(tr, res.h, 0, file_len, ctypes.byref(out))
        if rc != SX_OK:
            raise SxError(rc, L.sx_transport_last_error(tr).decode())
        if out.value:
            whole = Result(scanner, out)
    if timings is not None:
        timings["scan_ms"] = 1e3 * (t1 - t0)
        timings["gather_ms"] = 1e3 * (time.perf_counter() - t1)
    return whole, res, counts


class ShardCounts(list):
    """Findings per rank; .overflow[k] = how many of rank k's last findings lie behind its range end."""
    overflow = None


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
    findings_bytes, arena_bytes = bytes(findings_bytes), bytes(arena_bytes)   # (numpy views of the gather buffer, or bytes)
    n = len(findings_bytes) // ctypes.sizeof(Finding)
    arr = (Finding * n).from_buffer_copy(findings_bytes)
    return [dict(position=f.position, precision=PRECISION[f.precision],
                 s=arena_bytes[f.str_off:f.str_off + f.str_len].decode("utf-8"), completes=bool(f.completes_previous),
                 mission_id=f.mission_id, file_id=f.input_file_id, slice_index=f.slice_index) for f in arr]
