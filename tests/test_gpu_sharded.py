"""GPU test of the sharded path with the real kernels: two/three gloo ranks share cuda:0, each
uploads only its own byte range (+halo) and calls sx_scan_shard_device; the gathered findings
must equal the oracle's sequential scan.  (The 8-GPU run uses the same code with backend nccl.)"""
import ctypes
import os

import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

import refconfig as rc
from test_sharded_gloo import CASES, _free_port, make_data, oracle_findings

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, kind, flags, halo, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import stringsext_amd as sx
        from stringsext_amd import sharded
        data = make_data(kind, 1234)
        ms = rc.missions(**flags)
        sc = sx.Scanner(ms, device=0)
        held = {}

        def get_buffer(lo, hi):
            if held.get("range") != (lo, hi):
                if "ptr" in held:
                    sc.free(held["ptr"])
                held["ptr"] = sc.alloc(hi - lo)
                sc.upload(held["ptr"], data[lo:hi])
                held["range"] = (lo, hi)
            return held["ptr"]

        gathered, _ = sharded.scan_sharded(sc, get_buffer, len(data), file_id=1, halo=halo, device="cpu")
        if rank == 0:
            parts = [sharded.decode_findings(fb, ab) for fb, ab in gathered]
            got = [(f["position"], f["precision"], f["s"], f["completes"], f["mission_id"], f["slice_index"])
                   for f in sharded.splice_order(parts, len(data))]
            want = oracle_findings(ms, data)
            q.put(("ok", got == want, len(got), len(want), next(((a, b) for a, b in zip(got, want) if a != b), None)))
    except Exception:  # pragma: no cover
        import traceback
        q.put(("err", traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,kind,flags,halo", CASES, ids=[f"{c[0]}ranks-{c[1]}-{i}" for i, c in enumerate(CASES)])
def test_sharded_device_scan_equals_sequential(world, kind, flags, halo):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, kind, flags, halo, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=900)
    for p in procs:
        p.join(timeout=120)
    assert res[0] == "ok", res[1]
    assert res[1], f"sharded != sequential: {res[2]} vs {res[3]} findings, first diff {res[4]}"


def _nccl_worker(port, q):
    """world size 1, backend nccl (= RCCL) on cuda:0: the exchange's all_gather_into_tensor and the gather's size table run on
    DEVICE tensors — the code the 8-GPU run takes, executed once"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        import stringsext_amd as sx
        from stringsext_amd import sharded
        data = make_data("c4", 1234)
        ms = rc.missions(encodings=["utf-8", "utf-16le", "utf-16be"], chars_min="10", unicode_block_filter="African")
        sc = sx.Scanner(ms, device=0)
        d = sc.alloc(len(data)); sc.upload(d, data)
        timings = {}
        gathered, res = sharded.scan_sharded(sc, lambda lo, hi: ctypes.c_void_p(d.value + lo), len(data), file_id=1, device="cuda:0",
                                             gather=True, timings=timings)
        parts = [sharded.decode_findings(fb, ab) for fb, ab in gathered]
        got = [(f["position"], f["precision"], f["s"], f["completes"], f["mission_id"], f["slice_index"])
               for f in sharded.splice_order(parts, len(data))]
        want = oracle_findings(ms, data)
        counts, _ = sharded.scan_sharded(sc, lambda lo, hi: ctypes.c_void_p(d.value + lo), len(data), file_id=1, device="cuda:0", gather=False)
        q.put(("ok", got == want and list(counts) == [len(want)], len(got), len(want), timings))
    except Exception:  # pragma: no cover
        import traceback
        q.put(("err", traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_the_nccl_transport_runs_once_on_one_gpu():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_worker, args=(_free_port(), q))
    p.start()
    res = q.get(timeout=600)
    p.join(timeout=120)
    assert res[0] == "ok", res[1]
    assert res[1], res


def _nccl2_worker(rank, port, q):
    """two ranks on two GPUs over RCCL: the exchange and the gather (batched point-to-point on both ends) as the 8-GPU run takes them"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=2, device_id=torch.device("cuda", rank))
    try:
        import stringsext_amd as sx
        from stringsext_amd import sharded
        data = make_data("c4", 4321)
        ms = rc.missions(encodings=["utf-8", "utf-16le", "utf-16be"], chars_min="10", unicode_block_filter="African")
        sc = sx.Scanner(ms, device=rank)
        d = sc.alloc(len(data)); sc.upload(d, data)
        gathered, res = sharded.scan_sharded(sc, lambda lo, hi: ctypes.c_void_p(d.value + lo), len(data), file_id=1, device=f"cuda:{rank}", gather=True)
        if rank == 0:
            parts = [sharded.decode_findings(fb, ab) for fb, ab in gathered]
            got = [(f["position"], f["precision"], f["s"], f["completes"], f["mission_id"], f["slice_index"]) for f in sharded.splice_order(parts, len(data))]
            want = oracle_findings(ms, data)
            q.put(("ok", got == want, len(got), len(want)))
    except Exception:  # pragma: no cover
        import traceback
        q.put(("err", traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_two_gpus_over_rccl():
    """skipped on a box with one GPU (the driver's test box); on a multi-GPU node: the gather over RCCL must not hang and must give the
    sequential scan's findings"""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_nccl2_worker, args=(r, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
    assert res[0] == "ok", res[1]
    assert res[1], res


def _library_worker(rank, world, port, q):
    """the LIBRARY's transport (csrc/sx_transport.cpp: RCCL through dlopen): sx_scan_sharded with sx_transport_allgather as its callback,
    sx_transport_gather + splice on rank 0; the torch.distributed group (gloo) only ships rank 0's 128-byte id"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import stringsext_amd as sx
        from stringsext_amd import sharded
        data = make_data("c4", 777)
        ms = rc.missions(encodings=["utf-8", "utf-16le", "utf-16be"], chars_min="10", unicode_block_filter="African")
        sc = sx.Scanner(ms, device=rank)
        d = sc.alloc(len(data)); sc.upload(d, data)
        timings = {}
        for _ in range(2):   # (twice: the transport and its buffers are kept from call to call)
            whole, res, counts = sharded.scan_sharded_library(sc, lambda lo, hi: ctypes.c_void_p(d.value + lo), len(data), rank, file_id=1, timings=timings)
        if rank == 0:
            got = [(f["position"], f["precision"], f["s"], f["completes"], f["mission_id"], f["slice_index"]) for f in whole.findings()]
            want = oracle_findings(ms, data)
            q.put(("ok", got == want and sum(counts) == len(want), len(got), len(want), timings))
    except Exception:  # pragma: no cover
        import traceback
        q.put(("err", traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def _run_library(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_library_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
    assert res[0] == "ok", res[1]
    assert res[1], res


def test_the_librarys_rccl_transport_runs_on_one_gpu():
    """world size 1: ncclCommInitRank, ncclAllGather (sx_scan_sharded's exchanges and the gather's size table) and the splice run once — the
    code a Rust host binds, no torch in the path"""
    _run_library(1)


def test_the_librarys_rccl_transport_on_two_gpus():
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    _run_library(2)


def test_bench_with_eight_ranks_on_one_gpu_through_gloo():
    """The driver's `bench.py --gpus N` line, N = 8, as far as one GPU can carry it: eight ranks sharing cuda:0, the exchange and the
    gather through gloo.  The default is STRONG scaling (BASELINE.json's metric: one image at 1/2/4/8 GPUs) — here a 1 GiB image —, the
    line says so, and every rank's shard went through the kernels."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--backend", "gloo", "--single-device", "--gib", "1",
                          "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 8 and line["scaling"] == "strong"
    assert line["config"]["image_bytes"] == 1 << 30 and line["config"]["bytes_per_gpu"] == (1 << 30) // 8
    assert "strong scaling" in line["config"]["workload"]
    assert line["gather_ms_per_step"] is not None and line["exchange_ms_per_step"] is not None
    assert line["findings_per_step"] > 0
