#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (`rocprofv3 --kernel-trace --stats -d DIR -o NAME -- cmd` writes
DIR/NAME_results.db) as the per-kernel table the round's profiles/ directory keeps.
usage: python tools/rocprof_summary.py gpurun_out/prof/r_results.db [> profiles/rNN_xxx.txt]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute(
    "select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3, "
    "max(vgpr_count), max(lds_size), max(scratch_size) from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows) or 1.0
print("%-64s %7s %12s %10s %10s %10s %6s %5s %7s %7s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "%", "vgpr", "lds", "scratch"))
for r in rows:
    name = r[0].replace("void apk::", "").split("(")[0]
    print("%-64s %7d %12.1f %10.1f %10.1f %10.1f %6.1f %5d %7d %7d" % (name[:64], r[1], r[2], r[3], r[4], r[5], 100 * r[2] / tot, r[6] or 0, r[7] or 0, r[8] or 0))
