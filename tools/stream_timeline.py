#!/usr/bin/env python3
"""Per-queue view of a rocprofv3 kernel trace (rocpd database) over the window of the run with the most kernel time in flight: for every hardware queue the
number of launches, the time with a kernel running and the idle time between kernels; for the busiest queue the launches one by
one.  Where do the milliseconds of a loaded stream go?
usage: python tools/stream_timeline.py db [window_ms = 150] [rows = 120]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
win = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 150e6
nrows = int(sys.argv[3]) if len(sys.argv) > 3 else 120
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = list(db.execute("select name, start, end, %s from kernels order by start" % (qcol or "0")))
# the window with the most kernel time in flight (the loaded region of a bench run, not its setup or its lone-proof legs)
buckets = {}
for r in rows:
    buckets[r[1] // int(win)] = buckets.get(r[1] // int(win), 0) + (r[2] - r[1])
mid = max(buckets, key=buckets.get) * int(win)
rows = [r for r in rows if mid <= r[1] < mid + win]
by_q = {}
for name, s, e, q in rows:
    by_q.setdefault(q, []).append((s, e, name))
# GPU-wide: time with at least one kernel running
ev = sorted([(s, 1) for _, s, e, _ in rows] + [(e, -1) for _, s, e, _ in rows])
depth, last, any_busy, area = 0, ev[0][0], 0, 0
for t, d in ev:
    if depth > 0:
        any_busy += t - last
    area += depth * (t - last)
    last = t
    depth += d
span = ev[-1][0] - ev[0][0]
print("column %s; window %.1f ms: %d launches on %d queues; some kernel running %.1f %% of the time, %.2f kernels in flight on average"
      % (qcol, span / 1e6, len(rows), len(by_q), 100.0 * any_busy / span, area / span))
print("%8s %8s %10s %10s %10s" % ("queue", "launches", "busy_ms", "idle_ms", "max_gap_us"))
best = None
for q, ks in sorted(by_q.items(), key=lambda kv: -len(kv[1])):
    busy = sum(e - s for s, e, _ in ks)
    gaps = [ks[i + 1][0] - ks[i][1] for i in range(len(ks) - 1)]
    idle = sum(g for g in gaps if g > 0)
    print("%8s %8d %10.2f %10.2f %10.1f" % (q, len(ks), busy / 1e6, idle / 1e6, (max(gaps) if gaps else 0) / 1e3))
    if best is None:
        best = ks
print("\nbusiest queue, launch by launch:")
print("%10s %9s %9s  %s" % ("start_us", "dur_us", "gap_us", "kernel"))
prev = best[0][0]
for s, e, name in best[:nrows]:
    name = name.replace("void apk::", "").split("(")[0]
    print("%10.1f %9.1f %9.1f  %s" % ((s - best[0][0]) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, name[:70]))
    prev = e

# the longest stretch with NO kernel running anywhere, and what ran on either side of it
ends = sorted((e, s, q, name) for name, s, e, q in rows)
starts = sorted((s, e, q, name) for name, s, e, q in rows)
run_end, best_gap = starts[0][1], (0, 0, 0)
for s, e, q, name in starts:
    if s > run_end and s - run_end > best_gap[0]:
        best_gap = (s - run_end, run_end, s)
    run_end = max(run_end, e)
print("\nlongest stretch with no kernel running anywhere: %.1f us" % (best_gap[0] / 1e3))
if best_gap[0]:
    g0, g1 = best_gap[1], best_gap[2]
    print("last launches to END before it:")
    for e, s, q, name in [x for x in ends if x[0] <= g0][-14:]:
        print("  queue %3s  end %10.1f us before the gap  dur %8.1f  %s" % (q, (g0 - e) / 1e3, (e - s) / 1e3, name.replace("void apk::", "").split("(")[0][:60]))
    print("first launches to START after it:")
    for s, e, q, name in [x for x in starts if x[0] >= g1][:14]:
        print("  queue %3s  start %8.1f us after the gap  dur %8.1f  %s" % (q, (s - g1) / 1e3, (e - s) / 1e3, name.replace("void apk::", "").split("(")[0][:60]))
