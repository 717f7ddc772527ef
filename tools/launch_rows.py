"""Every scan / wave launch of one bench run, by where it ran: warm-up, timed steps, alone — from a rocprofv3 --kernel-trace CSV.
bench.py launches a 16-byte fill_kernel right before and right after its timed region; the launches between those two markers are
the K timed steps.  Prints one row per kernel and class (launches, average / min / max ms, GB/s at the launch's algorithmic bytes)
and the roofline fraction recomputed from the timed rows alone: the figure bench.py prints must come out of this file.
usage: tools/launch_rows.py KERNEL_TRACE.csv BENCH_LINE.json > profiles/TAG_launch_rows.csv"""
import csv, json, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
bench = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
nbytes = bench["config"]["bytes_per_gpu"]
# a buffer whose output is gigabytes is scanned in pieces (bench.py "sequential_pieces"): a timed launch then covers one piece
pieces = max(1, int(bench.get("sequential_pieces") or 0))
# ... and a large shard with a fused scan in two halves (bench.py roofline.fused.launches_per_step): a fused launch then covers one half
fused_pieces = max(1, int(((bench.get("roofline") or {}).get("fused") or {}).get("launches_per_step") or 1))
fills = [i for i, r in enumerate(rows) if "fill_kernel" in r["Kernel_Name"]]
# the markers are the last two fill launches (the first ones generate the input)
m0, m1 = fills[-2], fills[-1]
def cls(i):
    return "warm-up" if i < m0 else "timed" if i < m1 else "alone"
import re
want = re.compile(r"\bsx::(scan_kernel(_dbcs|_fused)?|wave_replay_kernel|wave_emit_kernel)<")
is_scan = lambda name: name.startswith("scan_kernel")
# Warm-up launches are listed one by one: that is where a Mission's first launch finds its record regions too small and is launched
# AGAIN with larger ones (the library then remembers the size: bench.py's "scan_kernels_launched_again" counts them per timed step).
# The "alone" launches too: 1 and 2 start on an idle chip (bench.py's per_kernel_ms_alone = the faster one), 3 and 4 are a pair queued
# back to back (SX_SCAN_WARM=1), 4 being per_kernel_ms_alone_behind_an_identical_launch.
agg = {}
seen_warm = {}
for i, r in enumerate(rows):
    n = r["Kernel_Name"]
    if not want.search(n):
        continue
    name = n.split("(")[0].replace("void ", "").replace("sx::", "")
    ms = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    c = cls(i)
    if c != "timed" and is_scan(name):
        seen_warm[(name, c)] = seen_warm.get((name, c), 0) + 1
        c = "%s launch %d" % (c, seen_warm[(name, c)])
    agg.setdefault((name, c), []).append(ms)
print("kernel,class,launches,avg_ms,min_ms,max_ms,gbs_at_avg")
tot_ms = tot_launch = tot_bytes = 0
for (name, c), v in sorted(agg.items()):
    avg = sum(v) / len(v)
    pcs = fused_pieces if name.startswith("scan_kernel_fused") else pieces
    per_launch_bytes = (nbytes / pcs if c == "timed" else nbytes) if is_scan(name) else nbytes / max(1, len(v) / max(1, bench["steps"]) ) if c == "timed" else nbytes
    print(f'"{name}",{c},{len(v)},{avg:.3f},{min(v):.3f},{max(v):.3f},{per_launch_bytes / (avg * 1e-3) / 1e9:.1f}')
    if c == "timed" and is_scan(name):
        tot_ms += sum(v); tot_launch += len(v); tot_bytes += len(v) * nbytes / pcs
if tot_launch:
    frac = tot_bytes / (tot_ms * 1e-3) / 1e9 / 8000.0
    print(f'# scan launches inside the timed region: {tot_launch} launches{" of a piece (1/%d of the buffer) each" % max(pieces, fused_pieces) if max(pieces, fused_pieces) > 1 else ""}, {tot_ms:.3f} ms -> {tot_bytes / (tot_ms * 1e-3) / 1e9:.1f} GB/s = {frac:.4f} of the 8 TB/s peak '
          f'(bench.py in the same run: {bench["roofline"]["frac"]}); a step = {tot_ms / bench["steps"]:.3f} ms of scan launches')
# string-dense Missions: their passes over the input are the wave kernels' count pass (MODE 0) and write pass (MODE 1); a pass of a
# single Mission is cut into slabs, i.e. several launches.  bench.py's figure for them comes from HIP events around each pass
# (the verify kernel and the two prefix sums between count and write are inside), so it is a little above these sums.
wave = {m: sum(sum(v) for (name, c), v in agg.items() if c == "timed" and name.startswith("wave_replay_kernel<%d," % m)) for m in (0, 1)}
wave[1] += sum(sum(v) for (name, c), v in agg.items() if c == "timed" and name.startswith("wave_emit_kernel<"))   # the lane-per-finding writer
if wave[0] or wave[1]:
    wp = bench["roofline"].get("wave_passes_ms", {})
    print(f'# wave kernels inside the timed region, per step: count pass {wave[0] / bench["steps"]:.3f} ms, write pass {wave[1] / bench["steps"]:.3f} ms '
          f'(bench.py in the same run, events around the passes: {wp.get("count")} / {wp.get("write")} ms for {wp.get("missions")} Mission(s))')
