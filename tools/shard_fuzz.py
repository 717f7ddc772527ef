"""CPU-only randomized check of the byte-range-sharded path (no GPU, no process group): the ranks of
stringsext_amd/sharded.py run one after the other in this process — own range + halo, start where the
predecessor stopped, wider halo when a run crosses it, splice_order at the end — with the host replay stage
fed by the oracle's runs, against the oracle's sequential scan.  Cases come from gpu_fuzz.py's generator.
usage: tools/shard_fuzz.py SECONDS [SEED]"""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fuzz_case
import stringsext_amd as sx
from stringsext_amd import sharded
from product_harness import oracle_runs_for_chunk
from test_sharded_gloo import oracle_findings

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
rng = random.Random(seed)
t0 = time.time(); n = 0
key = lambda f: (f["position"], f["precision"], f["s"], f["completes"], f["mission_id"], f["slice_index"])
while time.time() - t0 < budget:
    case_seed = rng.randrange(1 << 31)
    c = fuzz_case.make(case_seed)
    data = c["files"][0][:400_000]
    if len(data) < 3 * 4096:
        continue
    ms = c["missions"]
    if any(m["chars_min_nb"] == 0 or m["encoding"] == 71 for m in ms):   # (71: ISO-2022-JP)
        continue   # refused by sx_scan_shard* (its stage B is one sequential pass)
    n += 1
    world = rng.choice([2, 3, 5])
    halo0 = rng.choice([4096, 8192, 65536])
    parts, prev_end = [], None
    for rank in range(world):
        own_lo, own_hi = sharded.shard_bounds(len(data), world, rank)
        sc = sx.Scanner(ms, device=sx.SX_HOST_ONLY)
        start = None if prev_end is None or all(p <= own_lo for p in prev_end) else [max(own_lo, p) for p in prev_end]
        h = halo0
        while True:
            buf_lo = max(0, own_lo - h) // 4096 * 4096
            buf_hi = min(len(data), own_hi + h)
            buf = data[buf_lo:buf_hi]
            try:
                res, ends = sc.scan_shard(buf, buf_lo, own_lo, own_hi, start_at=start, file_id=1,
                                          runs_per_mission=oracle_runs_for_chunk(ms, buf, buf_lo))
            except sx.SxError as e:
                if e.code != sx.SX_E_HALO or buf_lo == 0:
                    raise
                h *= 8
                continue
            if any(e >= buf_hi for e in ends) and buf_hi < len(data):
                res.free(); h *= 8
                continue
            break
        if prev_end is not None:
            ends = [max(e, p) for e, p in zip(ends, prev_end)]
        parts.append(res.findings()); res.free(); sc.close()
        prev_end = ends
    got = [key(f) for f in sharded.splice_order(parts, len(data))]
    want = oracle_findings(ms, data)
    if got != want:
        print(f"MISMATCH shard fuzz seed {seed} case_seed {case_seed} world={world} halo={halo0}: {fuzz_case.describe(c)}")
        print("  first diff", next(((a, b) for a, b in zip(got, want) if a != b), (len(got), len(want))))
        sys.exit(1)
print(f"shard fuzz seed {seed}: {n} cases, all equal to the oracle")
