"""profiles/traffic.json from two counter passes (tools/pmc_pass.sh fetch "FETCH_SIZE", tools/pmc_pass.sh write "WRITE_SIZE"):
HBM bytes per scan launch of the headline bench.  FETCH_SIZE and WRITE_SIZE count KiB; on gfx950 wide coalesced reads are
tallied at half their bytes (MI355X_MICROARCH.md, HBM section) -> FETCH_SIZE x 2.  usage: tools/make_traffic.py ROUND_TAG"""
import csv, json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
tag = sys.argv[1]
# Round 6: the counter passes run `bench.py --steps 1 --warmup 0 --no-alone`: ONE pass over the 64 GiB — the fused launch's halves —
# plus whatever launches of the per-Mission kernels the step needed; the sum over a kernel's dispatches is its traffic per pass.
def per_launch(path, counter):
    out = {}
    for r in csv.DictReader(open(path)):
        if "::scan_kernel" in r["kernel"] and r["counter"] == counter:
            out[r["kernel"].replace("void ", "").replace("sx::", "")] = float(r["sum_over_dispatches"])
    return out
fetch = per_launch(os.path.join(ROOT, "profiles", f"{tag}_fetch.csv"), "FETCH_SIZE")
write = per_launch(os.path.join(ROOT, "profiles", f"{tag}_write.csv"), "WRITE_SIZE")
assert fetch and set(fetch) == set(write), (fetch, write)
per_kernel = {k: (fetch[k] * 2 + write[k]) * 1024 for k in fetch}
tj = {"workload": "c3", "bytes_per_gpu": 64 << 30, "scan_kernel_source": bench.scan_kernel_source_hash(),
      "traffic_bytes_per_launch": int(sum(per_kernel.values()) / len(per_kernel)),   # (per pass over the shard: the fused launch's halves together)
      "method": f"rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-alone` "
                f"(tools/pmc_pass.sh; profiles/{tag}_fetch.csv, {tag}_write.csv: sums over the dispatches of each kernel); "
                "counters are in KiB; FETCH_SIZE x2 on gfx950 (wide coalesced reads are tallied at half their bytes, MI355X_MICROARCH.md), WRITE_SIZE as is "
                "(calibration: fill_kernel's WRITE_SIZE is exactly the 64 GiB it writes); round 6: the sum over the dispatches of the fused scan kernel in that one step = one pass over the shard",
      "fetch_kib_per_launch": fetch, "write_kib_per_launch": write, "bytes_per_launch_per_kernel": per_kernel}
# the fused kernel's vector-instruction issue, if the SQ passes of the same session are there (tools/pmc_pass.sh TAG_sq / TAG_sq2, --no-alone):
# instructions per 1 KiB tile, and the share of the chip's cycles in which every SIMD issues one (a wave64 vector instruction takes a SIMD four
# cycles: SQ_ACTIVE_INST_VALU counts quad-cycles; 1024 SIMDs; GRBM_GUI_ACTIVE sums the 8 XCDs)
try:
    sq = {}
    for name in (f"{tag}_pmc_c3_sq.csv", f"{tag}_pmc_c3_sq2.csv"):
        for r in csv.DictReader(open(os.path.join(ROOT, "profiles", name))):
            if "scan_kernel_fused" in r["kernel"]:
                sq[r["counter"]] = float(r["sum_over_dispatches"])
    tiles = (64 << 30) / 1024
    tj["valu"] = {"insts_per_tile": round(sq["SQ_INSTS_VALU"] / tiles, 1), "salu_per_tile": round(sq["SQ_INSTS_SALU"] / tiles, 1),
                  "issue_fraction": round(sq["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (sq["GRBM_GUI_ACTIVE"] / 8), 3),
                  "source": f"profiles/{tag}_pmc_c3_sq.csv, {tag}_pmc_c3_sq2.csv (one pass over 64 GiB each)"}
except (OSError, KeyError):
    pass
json.dump(tj, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
print(json.dumps(tj, indent=1))
