#!/usr/bin/env python3
"""rocprofv3 (rocpd sqlite output) -> per-kernel stats CSV, like `--stats` prints.
usage: tools/rocpd_stats.py gpurun_out/prof_x/x_results.db > profiles/x_kernel_stats.csv"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       "from kernels group by name order by 3 desc"))
total = sum(r[2] for r in rows) or 1
print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"')
for name, calls, tot, avg, mn, mx in rows:
    print(f'"{name}",{calls},{tot},{avg:.1f},{100.0 * tot / total:.3f},{mn},{mx}')
