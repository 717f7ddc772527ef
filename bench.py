#!/usr/bin/env python3
"""bench.py — GiB/s scanned on BASELINE.json's headline workload (C3(i): 64 GiB synthetic
background, `-e utf-8 -e utf-16le -e utf-16be -n 10 -u African`), one process per GPU.

A step = one pass of the hot path (sx_scan_device: ONE fused HIP scan kernel that reads the shard once for
the three Missions — round 6; `--per-mission-launches`: one kernel per Mission, three reads, rounds 1-5 —,
records packed and joined on the device, exact replay of the regions around long runs on the device ->
findings in reference order in pinned host memory) over the rank's HBM-resident shard.
Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline`
(HIP-event kernel time vs the 8 TB/s HBM peak) and `cpu_baseline` (the oracle, the only
runnable restatement of the reference, on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
SEED = 0x5EED5EED5EED5EED

WORKLOADS = {
    # BASELINE.json configs[2]: the configuration the metric is quoted on
    "c3": dict(flags=dict(encodings=["utf-8", "utf-16le", "utf-16be"], chars_min="10",
                          unicode_block_filter="African"), gib=64.0,
               name="C3(i): -e utf-8 -e utf-16le -e utf-16be -n 10 -u African -t x, synthetic background"),
    # BASELINE.json configs[3]: the same three Missions on a 256 GiB image, byte-range sharded over 8 GPUs = 32 GiB per rank
    # (`--gpus 8 --workload c4` is the stated configuration; with fewer ranks it is the same per-rank share of a smaller image)
    "c4": dict(flags=dict(encodings=["utf-8", "utf-16le", "utf-16be"], chars_min="10",
                          unicode_block_filter="African"), gib=32.0,
               name="C4: -e utf-8 -e utf-16le -e utf-16be -n 10 -u African -t x, synthetic background, 32 GiB per rank (256 GiB over 8 ranks), "
                    "byte-range shards + halo, Finding buffers gathered to rank 0"),
    # BASELINE.json configs[0] at a GPU-sized length (the text-dense extreme: ~11.8 k findings per MiB)
    "c1": dict(flags=dict(encodings=["ascii"], chars_min="4"), gib=1.0, kernels="SingleByteRange",
               name="C1-like: -e ascii -n 4 -t x, synthetic background (dense: every 85th byte starts a finding)"),
    # BASELINE.json configs[1]
    "c2": dict(flags=dict(encodings=["utf-8"], chars_min="10"), gib=4.0, kernels="Utf8Range2",
               name="C2: -e utf-8 -n 10 -t x, synthetic background"),
    # BASELINE.json configs[4]: six missions, the legacy ones with the per-encoding filters SURVEY.md 8(a) recommends
    # (with a global -u African no CJK or Cyrillic character could ever pass and the tables would never matter)
    "c5": dict(flags=dict(encodings=["utf-8,,,African", "utf-16le,,,African", "utf-16be,,,African", "big5,,,Cjk",
                                     "euc-jp,,,Asian", "koi8-r,,,Cyrillic"], chars_min="10"), gib=64.0,
               kernels="Utf8Range2|Utf16Range x2|scan_kernel_dbcs<4> (Big5)|scan_kernel_dbcs<5> (EUC-JP)|SingleByteLut (KOI8-R)",
               name="C5: -e utf-8,,,African -e utf-16le,,,African -e utf-16be,,,African -e big5,,,Cjk -e euc-jp,,,Asian "
                    "-e koi8-r,,,Cyrillic -n 10 -t x, synthetic background"),
}


def scan_kernel_source_hash():
    """What roofline.traffic was measured for: the PMC passes are separate runs (profiles/traffic.json), valid only
    as long as the scan kernels' source is the one they ran."""
    import hashlib
    h = hashlib.sha256()
    for f in ("sx_kernels.hip", "sx_fused.hip", "sx_scan_core.hpp", "sx_classify_ranges.hpp", "sx_device.hpp"):
        h.update(open(os.path.join(ROOT, "stringsext_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=2)   # (two: the first pass learns which Missions are string-dense, the second sizes the pools of the path they then take)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--gib", type=float, default=None, help="bytes per GPU in GiB (default: the workload's size)")
    ap.add_argument("--subchunk-kib", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-mib", type=int, default=512)
    ap.add_argument("--background", default="random", choices=["random", "zero"],
                    help="zero: a constant-byte buffer instead of the synthetic background (counter passes: same instructions, other data)")
    ap.add_argument("--backend", default="nccl", help="process-group backend (testing the N>1 path on one GPU: gloo)")
    ap.add_argument("--single-device", action="store_true", help="testing: every rank uses cuda:0")
    ap.add_argument("--transport", default="torch", choices=["torch", "library"],
                    help="N > 1: the exchange and the gather through torch.distributed (default) or through the library's own RCCL transport "
                         "(csrc/sx_transport.cpp: sx_transport_allgather / sx_transport_gather; the process group then only ships rank 0's 128-byte id)")
    ap.add_argument("--result-on-device", action="store_true",
                    help="SX_OPT_RESULT_ON_DEVICE: a single-Mission workload's dense result stays in HBM (c1) — NOT the headline boundary, the line's "
                         "config says so")
    ap.add_argument("--per-mission-launches", action="store_true",
                    help="SX_OPT_NO_FUSED_SCAN: one scan launch per Mission, each reading the whole buffer (rounds 1-5; the line's config.passes says so) "
                         "instead of the fused launch that reads it once")
    ap.add_argument("--no-alone", action="store_true", help="skip the launches outside the timed region (counter passes: the step's launches only)")
    ap.add_argument("--generic-kernels", action="store_true",
                    help="force the table-driven (LUT) classifiers instead of the range kernels: what a Mission with an arbitrary af / ubf costs")
    ap.add_argument("--ubf", default=None,
                    help="another -u for the workload's Missions (what an alias filter costs: --ubf Cjk, --ubf Asian, --ubf All); the line's "
                         "config.workload says so — not the headline configuration")
    ap.add_argument("--chars-min", default=None,
                    help="another -n for the workload's Missions (the reference's default is 4); the line's config.workload says so — not the headline configuration")
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"],
                    help="strong: the workload's ONE image is split over the ranks — the default for `--gpus N` with N > 1 (BASELINE.json's "
                         "metric: 64 GiB at 1/2/4/8 GPUs); weak: every rank scans the workload's size — the default of `--workload c4` "
                         "(BASELINE config 4: 32 GiB per rank, 256 GiB over eight) and of N = 1")
    args = ap.parse_args()
    if args.scaling is None:   # (round 5: the driver's `--gpus N` line is on the metric's configuration; round 4 defaulted to N x 64 GiB)
        args.scaling = "strong" if args.gpus > 1 and args.workload != "c4" else "weak"

    # `python bench.py --gpus N` without a launcher: start the N ranks the way the driver does (one process per GPU over
    # torch.distributed.run on 127.0.0.1) and pass their output through
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import subprocess
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        # (--standalone: torchrun picks the rendezvous port itself — a port found by bind-and-close here could be taken before it is used)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=env))

    import torch
    import torch.distributed as dist

    import refconfig as rc
    import stringsext_amd as sx

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the scan has no CPU path")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    xdev = f"cuda:{local_rank}" if args.backend == "nccl" else "cpu"   # where the exchanged tensors live
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    wl = WORKLOADS[args.workload]
    if args.ubf:
        wl = dict(wl, flags=dict(wl["flags"], unicode_block_filter=args.ubf), name=wl["name"].replace("-u African", "-u " + args.ubf) + f" [--ubf {args.ubf}]",
                  kernels="the classifiers of -u %s: Utf8Range3T|Utf16RangesT x2 for an alias filter of three-byte leads, the table kernels for All" % args.ubf)
    if args.chars_min:
        wl = dict(wl, flags=dict(wl["flags"], chars_min=args.chars_min), name=wl["name"].replace("-n %s" % wl["flags"]["chars_min"], "-n " + args.chars_min) + f" [--chars-min {args.chars_min}]")
    missions = sx.missions_from_flags(**wl["flags"])   # the product's front end, from the literal flag strings
    assert missions == rc.missions(**wl["flags"])      # (the reference's rules restated in tests/refconfig.py: the checker)
    nbytes = int((args.gib if args.gib is not None else wl["gib"]) * (1 << 30)) // 4096 * 4096
    if args.scaling == "strong":   # ONE image of the workload's size, a byte range of it per rank
        nbytes = nbytes // world // 4096 * 4096
    sc = sx.Scanner(missions, device=local_rank, subchunk_bytes=args.subchunk_kib * 1024, generic_kernels=args.generic_kernels,
                    result_on_device=args.result_on_device, fused_scan=not args.per_mission_launches)

    # rank r owns bytes [r*nbytes, (r+1)*nbytes) of ONE world*nbytes image (weak scaling: nbytes = the workload's size; strong: its
    # N-th part); for N > 1 its buffer also holds a halo on both sides (runs that cross a shard boundary)
    import ctypes
    from stringsext_amd import sharded
    file_len = world * nbytes
    halo = sharded.HALO_DEFAULT if world > 1 else 0
    own_lo, own_hi = rank * nbytes, (rank + 1) * nbytes
    state = {}

    def get_buffer(lo, hi):
        if state.get("range") != (lo, hi):
            state["buf"] = None
            state["buf"] = torch.empty(hi - lo, dtype=torch.uint8, device=f"cuda:{local_rank}")
            if args.background == "zero":
                state["buf"].zero_()
            else:
                sc.fill_background(ctypes.c_void_p(state["buf"].data_ptr()), lo, hi - lo, SEED)
            torch.cuda.synchronize()
            state["range"] = (lo, hi)
        return ctypes.c_void_p(state["buf"].data_ptr())

    dptr = get_buffer(max(0, own_lo - halo) // 4096 * 4096, min(file_len, own_hi + halo))

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    timings = {}

    def step():
        sc.reset()
        if world == 1:
            res = sc.scan_device(dptr, nbytes, file_id=1)
            n = len(res)
        else:
            # shard scan + one all_gather over RCCL ("where did everybody start and stop", finding counts), then the
            # gather of the Finding buffers to rank 0 (BASELINE config 4) — all inside the timed region
            if args.transport == "library":
                whole, res, _counts = sharded.scan_sharded_library(sc, get_buffer, file_len, local_rank, file_id=1, halo=halo, timings=timings)
                n = len(whole) if whole is not None else 0
                if whole is not None:
                    whole.free()
            else:
                gathered, res = sharded.scan_sharded(sc, get_buffer, file_len, file_id=1, halo=halo,
                                                     device=xdev, gather=True, timings=timings)
                n = sum(len(fb) // 32 for fb, _ in gathered) if rank == 0 else 0
        st = sc.stats()
        res.free()
        return n, st

    # a 16-byte fill_kernel launch right before and right after the timed region: in a rocprofv3 --kernel-trace of this command the
    # scan launches between the two markers are the K timed steps, those before them the warm-up (with its re-scans), those after
    # them the "alone" launches (tools/launch_rows.py splits the trace there)
    marker_buf = torch.empty(4096, dtype=torch.uint8, device=f"cuda:{local_rank}")

    def marker():
        sc.fill_background(ctypes.c_void_p(marker_buf.data_ptr()), 0, 16, SEED)
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    marker()
    barrier()
    t0 = time.perf_counter()
    kernel_ms = [0.0] * len(missions)
    fused_ms, fused_mask, fused_launches = 0.0, 0, 0
    device_ms = replay_ms = d2h_ms = wave_count_ms = wave_write_ms = 0.0
    wave_windows = rescans = seq_pieces = fast_regions = general_regions = 0
    rescan_ms = 0.0
    findings = records = replay_bytes = 0
    for _ in range(args.steps):
        n, st = step()
        findings = n
        records = st.run_records
        replay_bytes = st.replay_bytes
        for k in range(len(missions)):
            kernel_ms[k] += st.kernel_ms[k]
        fused_ms += st.fused_ms
        fused_launches += st.fused_launches
        fused_mask = st.fused_mask
        device_ms += st.device_ms
        rescans += st.rescans
        rescan_ms += st.rescan_ms
        replay_ms += st.replay_ms
        d2h_ms += st.d2h_ms
        wave_count_ms += st.wave_count_ms
        wave_write_ms += st.wave_write_ms
        wave_windows = st.wave_windows
        seq_pieces = st.seq_pieces
        fast_regions, general_regions = st.fast_regions, st.general_regions
    barrier()
    dt = time.perf_counter() - t0
    marker()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=xdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # outside the timed region: every mission's scan kernel launched alone (nothing else on the
    # device), for the roofline's "what the kernel can do" next to "what it did in the job"
    alone_ms = []
    alone_warm_ms = []
    fused_alone_ms = None
    if rank == 0 and not args.no_alone:
        for k, m in enumerate(missions):
            mc = max(1, min(m["chars_min_nb"], m["output_line_char_nb_max"]))
            best = None
            for _ in range(2):
                sc.device_runs(k, dptr, nbytes, stream_parity=0, min_chars=mc, count_only=True)
                t = sc.stats().kernel_ms[k]
                best = t if best is None else min(best, t)
            alone_ms.append(best)
        if fused_mask:   # the fused launch alone: stage A of its Missions in one call
            idx = [k for k in range(len(missions)) if (fused_mask >> k) & 1]
            mcs = [max(1, min(missions[k]["chars_min_nb"], missions[k]["output_line_char_nb_max"])) for k in idx]
            for _ in range(3):
                sc.device_runs_multi(idx, dptr, nbytes, stream_parity=0, min_chars=mcs)
                t = sc.stats().fused_ms
                fused_alone_ms = t if fused_alone_ms is None else min(fused_alone_ms, t)
        # the same launch right behind an identical one (SX_SCAN_WARM): the chip is busy when it starts, as inside the job
        os.environ["SX_SCAN_WARM"] = "1"
        for k, m in enumerate(missions):
            mc = max(1, min(m["chars_min_nb"], m["output_line_char_nb_max"]))
            sc.device_runs(k, dptr, nbytes, stream_parity=0, min_chars=mc, count_only=True)
            alone_warm_ms.append(sc.stats().kernel_ms[k])
        os.environ.pop("SX_SCAN_WARM", None)

    K = max(args.steps, 1)
    kernel_ms = [x / K for x in kernel_ms]
    fused_ms /= K
    device_ms /= K
    out = None
    if rank == 0:
        total_bytes = world * nbytes * K
        value = total_bytes / dt / (1 << 30)
        # Dominant kernel = the scan kernels, one launch per mission per step, each reading the
        # whole shard once (algorithmic bytes per launch = nbytes, SURVEY.md §8d: 1 byte per
        # input byte x Mission pass; writes ~0).  The library queues them in ONE stream, so the
        # roofline figure is the average over the launches: (missions x nbytes) / sum of the
        # launch durations, HIP events around every launch on the scan stream.  (With
        # SX_MISSION_STREAMS=1 they overlap on a stream each and the span is the longest one.)
        # A Mission whose last buffer was string-dense has no scan kernel: its stage B replays every window with the
        # wave-cooperative kernels (sx_wave_dev.hip), a count pass and a write pass that each read the shard once — they are the
        # Mission's passes over the input and enter the average as such (their durations: HIP events around their launches).
        # Round 6: the Missions of fused_mask share ONE launch that reads the shard once (sx_fused.hip): one pass, its duration counted once
        # (stats.fused_ms; kernel_ms[k] of each of them repeats it).  SURVEY.md 8(d): passes = 1 for Missions fused into one read.
        fused_ks = [k for k in range(len(missions)) if (fused_mask >> k) & 1 and kernel_ms[k] > 0]
        own_ms = [0.0 if k in fused_ks else kernel_ms[k] for k in range(len(missions))]
        n_scanned = sum(1 for x in own_ms if x > 0) + (1 if fused_ks else 0)
        n_missions_scanned = sum(1 for x in kernel_ms if x > 0)
        n_wave = int(round(wave_windows / (nbytes / 128.0))) if wave_windows else 0
        wave_count_ms /= K
        wave_write_ms /= K
        span_ms = max(kernel_ms) if os.environ.get("SX_MISSION_STREAMS") else sum(own_ms) + (fused_ms if fused_ks else 0.0)
        passes = n_scanned + (2 * n_wave if (n_wave and n_missions_scanned < len(missions)) else 0)
        span_all = span_ms + ((wave_count_ms + wave_write_ms) if n_missions_scanned < len(missions) else 0.0)
        agg_gbs = passes * nbytes / (span_all * 1e-3) / 1e9
        # HBM bytes per launch from the PMC counters: they cannot be read inside this process, so the
        # value comes from the committed counter passes of this very command line (profiles/traffic.json
        # says how); null for any other workload or size
        traffic = valu = None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            if tj["workload"] == args.workload and tj["bytes_per_gpu"] == nbytes and tj.get("scan_kernel_source") == scan_kernel_source_hash():
                traffic = tj["traffic_bytes_per_launch"]   # else null: the counters were taken with other kernels
                valu = tj.get("valu")
        except (OSError, KeyError, ValueError):
            pass
        roofline = {
            "bound": "hbm", "achieved": round(agg_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(agg_gbs / HBM_PEAK_GBS, 4), "traffic": traffic,
            # what bounds the fused kernel instead of HBM: vector-instruction issue (separate SQ counter passes of this command, like `traffic`)
            "valu_issue": valu,
            "kernel": (("sx::scan_kernel_fused: ONE launch reads the shard once for Missions %s" % fused_ks
                        + ("; + sx::scan_kernel, %d launch(es) of the other Missions, average over all launches" % (n_scanned - 1) if n_scanned > 1 else ""))
                       if fused_ks else "sx::scan_kernel<%s>, %d launches per step, average" % (wl.get("kernels", "Utf8Range2|Utf16Range"), n_scanned))
                      + (" + sx::wave_replay_kernel count and write pass of %d string-dense Mission(s) (no scan kernel: every window is replayed)" % n_wave
                         if n_missions_scanned < len(missions) and n_wave else ""),
            "wave_passes_ms": {"missions": n_wave, "count": round(wave_count_ms, 3), "write": round(wave_write_ms, 3),
                               "count_gbs": round(n_wave * nbytes / (wave_count_ms * 1e-3) / 1e9, 1) if wave_count_ms > 0 else None,
                               "write_gbs": round(n_wave * nbytes / (wave_write_ms * 1e-3) / 1e9, 1) if wave_write_ms > 0 else None},
            "algorithmic_bytes_per_launch": nbytes,
            "fused": {"missions": fused_ks, "ms": round(fused_ms, 3), "gbs": round(nbytes / (fused_ms * 1e-3) / 1e9, 1) if fused_ms > 0 else None,
                      "launches_per_step": fused_launches // K,   # (a shard of >= 16 GiB: two halves)
                      "ms_alone": round(fused_alone_ms, 3) if rank == 0 and fused_alone_ms else None} if fused_ks else None,
            "per_kernel_ms": [round(x, 3) for x in kernel_ms],
            "per_kernel_gbs": [round(nbytes / (x * 1e-3) / 1e9, 1) if x > 0 else None for x in kernel_ms],
            "note": ("durations inside the timed region; round 6: the fused launch reads the shard ONCE for its Missions (frac = 1 x bytes / its time, SURVEY 8(d)); a shard of >= 16 GiB is scanned "
                     "in two halves, the first half's stage B and copy to the host under the second half's scan (fused.ms = both launches)") if fused_ks else
                    "durations inside the timed region; one launch per Mission: the busiest mission is scanned first and its stage B runs next to the other missions' scan launches",
            "per_kernel_ms_alone": [round(x, 3) for x in alone_ms],
            "per_kernel_ms_alone_behind_an_identical_launch": [round(x, 3) for x in alone_warm_ms],
            # (the per-Mission kernels, one launch each over the whole shard: what a Mission costs when it is not fused)
            "frac_alone": round(len(missions) * nbytes / (sum(alone_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if alone_ms and sum(alone_ms) > 0 else None,
            # the same passes over the WALL time of a step (stage B, merger and copies included): what "frac" cannot say when
            # kernels of several Missions run next to each other (their durations overlap and are counted in full above)
            "frac_of_step_wall": round(passes * nbytes / (dt / K) / 1e9 / HBM_PEAK_GBS, 4),
        }
        out = {
            "metric": "GiB/s scanned", "value": round(value, 2), "unit": "GiB/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / K * 1e3, 3),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "u8", "data": "synthetic" if args.background == "random" else "constant bytes (counter pass)",
            "config": {"workload": wl["name"] + (f" — ONE image of {world * nbytes / 2**30:g} GiB split over {world} ranks (strong scaling)" if world > 1 and args.scaling == "strong"
                                                  else f" — {nbytes / 2**30:g} GiB per rank, {world * nbytes / 2**30:g} GiB in all (weak scaling)" if world > 1 else ""),
                       "bytes_per_gpu": nbytes, "missions": len(missions),
                       "image_bytes": world * nbytes,
                       "parallelism": f"byte-range shards x{world}" + (" on ONE device (--single-device: plumbing test)" if args.single_device and world > 1 else ""),
                       "backend": args.backend if world > 1 else None,
                       "transport": (args.transport if world > 1 else None),
                       # the Missions' scan launches queue up in ONE HIP stream by default (measured: a stream per Mission — SX_OPT_MISSION_STREAMS,
                       # the north star's wording — aliases onto the same hardware queues and the kernels are bound by issue, not by launch order)
                       "mission_streams": "per mission" if os.environ.get("SX_MISSION_STREAMS") else "one scan stream + one stage-B stream",
                       "scan": ("fused: one launch, one read of the shard for Missions %s" % fused_ks) if fused_ks else "one launch per Mission, each reads the shard",
                       "records": "sx_finding16 (16 B) for string-dense segments, sx_finding (32 B) else",
                       **({"result": "left in HBM (SX_OPT_RESULT_ON_DEVICE): the step ends when the writer is done, no copy to the host"} if args.result_on_device else {}),
                       "passes": passes},
            "roofline": roofline,
            "breakdown_ms_per_step": {"scan_kernels_sum": round(span_ms, 3),
                                      "host_waits_for_stage_a": round(device_ms, 3),
                                      "scan_kernels_launched_again": {"launches": round(rescans / K, 3), "ms": round(rescan_ms / K, 3)},
                                      "sparse_download_for_host_replay": round(d2h_ms / K, 3),
                                      "host_part_of_stage_b": round(replay_ms / K, 3)},
            "gather_ms_per_step": round(timings.get("gather_ms", 0.0), 3) if world > 1 else None,
            "exchange_ms_per_step": round(timings.get("exchange_ms", 0.0), 3) if world > 1 else None,
            "findings_per_step": findings, "run_records_rank0": records,
            # stage B, lane per region (round 5): regions the fast pre-pass settled / left to the general replay kernel (all steps)
            "stage_b_regions": {"fast_prepass": fast_regions, "general_kernel": general_regions},
            "replay_fraction": round(replay_bytes / (len(missions) * nbytes), 5),
            # a buffer whose output is gigabytes (several string-dense Missions) is scanned in this many pieces, one after the other,
            # the copy of a piece's interleaved findings next to the following piece's kernels (0: in one go)
            "sequential_pieces": int(seq_pieces),
            "kernels_only_gib_s": round(world * nbytes / (span_all * 1e-3) / (1 << 30), 1) if span_all > 0 else None,
        }
        if not args.no_cpu_baseline:
            import threading
            import sxo_binding as sxo

            def usable_cores():
                n = len(os.sched_getaffinity(0))
                try:   # the cgroup quota can be far below the visible cores
                    q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
                    if q != "max":
                        n = min(n, max(1, -(-int(q) // int(per))))
                except (OSError, ValueError):
                    pass
                return n

            def timed(jobs):
                th = [threading.Thread(target=j) for j in jobs]
                t1 = time.perf_counter()
                for t in th:
                    t.start()
                for t in th:
                    t.join()
                return time.perf_counter() - t1   # ctypes drops the GIL inside the C oracle

            # (1) the reference's own threading model: one worker per Mission over the whole input
            #     (src/main.rs:97-151; the merger thread only interleaves)
            sample = min(args.cpu_sample_mib << 20, nbytes)
            host = sxo.background(0, sample, SEED)
            counts = [0] * len(missions)

            def per_mission(k):
                def run():
                    counts[k] = sxo.run_count([missions[k]], [host])[0]
                return run
            dt_ref = timed([per_mission(k) for k in range(len(missions))])
            # (2) byte-range sharded over every core this process may use: each thread scans
            #     its own range with all missions; the splice at the range edges is left out (throughput only)
            cores = usable_cores()
            part = min(sample, (128 << 20)) // 4096 * 4096
            parts = [sxo.background(c * part, part, SEED) for c in range(cores)]
            found = [0] * cores

            def per_range(c):
                def run():
                    found[c] = sxo.run_count(missions, [parts[c]])[0]
                return run
            dt_sh = timed([per_range(c) for c in range(cores)])
            out["cpu_baseline"] = {
                "value": round(cores * part / dt_sh / (1 << 30), 4), "unit": "GiB/s", "cores": cores, "kind": "port",
                "sample": f"{cores} byte ranges of {part >> 20} MiB of the same background, one thread each, all "
                          f"{len(missions)} missions per range, oracle/libsxo.so (C restatement, -O3); {sum(found)} findings",
                "reference_threading_model": {
                    "value": round(sample / dt_ref / (1 << 30), 4), "unit": "GiB/s", "cores": len(missions),
                    "sample": f"first {sample >> 20} MiB, one thread per mission as in src/main.rs:97-151; {sum(counts)} findings"},
            }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()   # rank 0 measures the kernels alone and the CPU baseline after the timed region: wait for it
    sc.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
