#!/usr/bin/env python3
"""Per-kernel sums of the PMC counters in a rocprofv3 rocpd database (`rocprofv3 --pmc A B -d DIR -o NAME -- cmd`).
usage: python tools/pmc_summary.py DIR/NAME_results.db"""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(set)
dur = collections.defaultdict(float)
seen = set()
for name, did, cname, val, d in db.execute("select kernel_name, dispatch_id, counter_name, value, duration from counters_collection"):
    k = name.replace("void apk::", "").split("(")[0][:56]
    acc[k][cname] += val
    calls[k].add(did)
    if did not in seen:
        seen.add(did)
        dur[k] += d / 1e3
cols = sorted({c for v in acc.values() for c in v})
tot = {c: sum(v[c] for v in acc.values()) or 1.0 for c in cols}
print("%-56s %6s %10s " % ("kernel", "calls", "total_us") + " ".join("%22s" % c for c in cols))
for k in sorted(acc, key=lambda k: -acc[k][cols[-1]]):
    print("%-56s %6d %10.1f " % (k, len(calls[k]), dur[k]) + " ".join("%14.4g (%4.1f%%)" % (acc[k][c], 100 * acc[k][c] / tot[c]) for c in cols))
