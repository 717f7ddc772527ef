"""Every scan / wave launch of one bench run, by where it ran: warm-up, timed steps, alone — from a rocprofv3 --kernel-trace CSV.
bench.py launches a 16-byte fill_kernel right before and right after its timed region; the launches between those two markers are
the K timed steps.  Prints one row per kernel and class (launches, average / min / max ms, GB/s at the launch's algorithmic bytes)
and the roofline fraction recomputed from the timed rows alone: the figure bench.py prints must come out of this file.
usage: tools/launch_rows.py KERNEL_TRACE.csv BENCH_LINE.json > profiles/TAG_launch_rows.csv"""
import csv, json, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
bench = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
nbytes = bench["config"]["bytes_per_gpu"]
fills = [i for i, r in enumerate(rows) if "fill_kernel" in r["Kernel_Name"]]
# the markers are the last two fill launches (the first ones generate the input)
m0, m1 = fills[-2], fills[-1]
def cls(i):
    return "warm-up" if i < m0 else "timed" if i < m1 else "alone"
want = ("scan_kernel", "wave_replay_kernel")
# a scan kernel launched AGAIN (its records did not fit: larger regions or the shared pool) runs on the library's second stream
import collections
scan_stream = collections.Counter(r.get("Stream_Id") for r in rows if "scan_kernel" in r["Kernel_Name"]).most_common(1)
scan_stream = scan_stream[0][0] if scan_stream else None
agg = {}
for i, r in enumerate(rows):
    n = r["Kernel_Name"]
    if not any(w in n for w in want):
        continue
    name = n.split("(")[0].replace("void ", "").replace("sx::", "")
    ms = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    c = cls(i)
    if "scan_kernel" in n and r.get("Stream_Id") != scan_stream:
        c += " (launched again)"
    agg.setdefault((name, c), []).append(ms)
print("kernel,class,launches,avg_ms,min_ms,max_ms,gbs_at_avg")
tot_ms = tot_launch = 0
for (name, c), v in sorted(agg.items()):
    avg = sum(v) / len(v)
    per_launch_bytes = nbytes if "scan_kernel" in name else nbytes / max(1, len(v) / max(1, bench["steps"]) ) if c == "timed" else nbytes
    print(f'"{name}",{c},{len(v)},{avg:.3f},{min(v):.3f},{max(v):.3f},{per_launch_bytes / (avg * 1e-3) / 1e9:.1f}')
    if c == "timed" and "scan_kernel" in name:
        tot_ms += sum(v); tot_launch += len(v)
if tot_launch:
    frac = tot_launch * nbytes / (tot_ms * 1e-3) / 1e9 / 8000.0
    print(f'# scan launches inside the timed region: {tot_launch} launches, {tot_ms:.3f} ms -> {tot_launch * nbytes / (tot_ms * 1e-3) / 1e9:.1f} GB/s = {frac:.4f} of the 8 TB/s peak '
          f'(bench.py in the same run: {bench["roofline"]["frac"]}); a step = {tot_ms / bench["steps"]:.3f} ms of scan launches')
# string-dense Missions: their passes over the input are the wave kernels' count pass (MODE 0) and write pass (MODE 1); a pass of a
# single Mission is cut into slabs, i.e. several launches.  bench.py's figure for them comes from HIP events around each pass
# (the verify kernel and the two prefix sums between count and write are inside), so it is a little above these sums.
wave = {m: sum(sum(v) for (name, c), v in agg.items() if c == "timed" and name.startswith("wave_replay_kernel<%d," % m)) for m in (0, 1)}
if wave[0] or wave[1]:
    wp = bench["roofline"].get("wave_passes_ms", {})
    print(f'# wave kernels inside the timed region, per step: count pass {wave[0] / bench["steps"]:.3f} ms, write pass {wave[1] / bench["steps"]:.3f} ms '
          f'(bench.py in the same run, events around the passes: {wp.get("count")} / {wp.get("write")} ms for {wp.get("missions")} Mission(s))')
