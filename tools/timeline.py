#!/usr/bin/env python3
"""Kernel timeline of the LAST `window_ms` of a rocprofv3 kernel trace (rocpd database): start offset, duration and the idle gap
before each launch - where a lone proof's milliseconds go between the kernels.
usage: python tools/timeline.py gpurun_out/x/r_results.db [window_ms]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
win = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 5e6
rows = list(db.execute("select name, start, end from kernels order by start"))
t_end = max(r[2] for r in rows)
rows = [r for r in rows if r[1] >= t_end - win]
t0 = rows[0][1]
prev_end = t0
busy = 0.0
gaps = 0.0
print("%10s %9s %9s  %s" % ("start_us", "dur_us", "gap_us", "kernel"))
for name, s, e in rows:
    gap = (s - prev_end) / 1e3
    name = name.replace("void apk::", "").split("(")[0]
    print("%10.1f %9.1f %9.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, name[:70]))
    busy += (e - s) / 1e3
    if gap > 0:
        gaps += gap
    prev_end = max(prev_end, e)
print("window %.1f us: kernels %.1f us, idle gaps %.1f us, %d launches" % ((prev_end - t0) / 1e3, busy, gaps, len(rows)))
