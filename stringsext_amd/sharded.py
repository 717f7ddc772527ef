"""Byte-range sharding of one input file over the GPUs of a node: one process per GPU,
`torch.distributed` for the two tiny exchanges the path really has —

  1. where each rank's replay stopped (one u64 per mission, chained rank by rank: a rank
     whose predecessor ran past the shard boundary repeats its replay from there), and
  2. the gather of the Finding buffers to rank 0 (RCCL over xGMI with backend "nccl").

There is no collective on the data path: every rank scans its own byte range (plus a halo)
with the same kernels as the single-GPU path.  The reference has nothing comparable (one
thread per Mission, src/main.rs:97-151); the splice reproduces what its single sequential
stream would have printed.
"""
import ctypes
import struct

import torch
import torch.distributed as dist

from . import Finding, PRECISION

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


def scan_sharded(scanner, get_buffer, file_len, file_id=1, file_stream_off=0, halo=HALO_DEFAULT, device="cpu",
                 runs_for_buffer=None, gather=True):
    """Scan one file of `file_len` bytes sharded over the process group.

    get_buffer(lo, hi) -> bytes | ctypes.c_void_p : the file bytes [lo, hi) on this rank.
    runs_for_buffer(buf_bytes, buf_off) -> runs per mission (tests on CPU: stage B only).
    Returns on rank 0 (gather=True): list of (findings_bytes, arena_bytes) per rank, in rank order;
    on other ranks None.  Every rank also gets its own Result as second value.
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
        res, ends = scanner.scan_shard(buf, buf_lo, own_lo, own_hi, start_at=start_at, file_stream_off=file_stream_off,
                                       file_id=file_id, reuse_runs=reuse, **kw)
        truncated = any(e >= buf_hi for e in ends) and buf_hi < file_len
        return res, ends, truncated, buf_hi

    # first attempt: everybody assumes the previous rank stops at the shard boundary
    h = halo
    res, ends, truncated, _ = attempt(None, h, False)
    while truncated:  # a run crosses the whole halo: look further
        h *= 8
        res.free()
        res, ends, truncated, _ = attempt(None, h, False)

    # chain: rank k learns where rank k-1 really stopped; almost always that is own_lo
    prev_end = [own_lo] * nm
    for k in range(1, world):
        src_ends = ends if rank == k - 1 else [0] * nm
        t = torch.tensor(src_ends, dtype=torch.int64, device=device)
        dist.broadcast(t, src=k - 1)
        if rank == k:
            prev_end = [int(x) for x in t.tolist()]
            if any(p > own_lo for p in prev_end):
                start = [max(own_lo, p) for p in prev_end]
                res.free()
                res, ends, truncated, _ = attempt(start, h, True)
                while truncated:
                    h *= 8
                    res.free()
                    res, ends, truncated, _ = attempt(start, h, False)
                ends = [max(e, p) for e, p in zip(ends, prev_end)]

    if not gather:
        return None, res
    fb, ab = res.raw()
    blob = struct.pack("<QQ", len(fb), len(ab)) + fb + ab
    sizes = _all_gather_u64([len(blob)], device)
    mx = max(s[0] for s in sizes)
    mine = torch.zeros(mx, dtype=torch.uint8, device=device)
    mine[:len(blob)] = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(device)
    if rank == 0:
        bufs = [torch.zeros(mx, dtype=torch.uint8, device=device) for _ in range(world)]
        dist.gather(mine, bufs, dst=0)
        out = []
        for b, s in zip(bufs, sizes):
            raw = bytes(b[:s[0]].cpu().numpy().tobytes())
            nf, na = struct.unpack("<QQ", raw[:16])
            out.append((raw[16:16 + nf], raw[16 + nf:16 + nf + na]))
        return out, res
    dist.gather(mine, None, dst=0)
    return None, res


def decode_findings(findings_bytes, arena_bytes):
    n = len(findings_bytes) // ctypes.sizeof(Finding)
    arr = (Finding * n).from_buffer_copy(findings_bytes)
    return [dict(position=f.position, precision=PRECISION[f.precision],
                 s=arena_bytes[f.str_off:f.str_off + f.str_len].decode("utf-8"), completes=bool(f.completes_previous),
                 mission_id=f.mission_id, file_id=f.input_file_id, slice_index=f.slice_index) for f in arr]
