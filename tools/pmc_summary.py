#!/usr/bin/env python3
"""Per-kernel sums of the PMC counters in a rocprofv3 rocpd database (`rocprofv3 --pmc A B -d DIR -o NAME -- cmd`).
usage: python tools/pmc_summary.py DIR/NAME_results.db [--json OUT.json [FACTS.json]]
--json also writes the table at full precision ({"kernels": {name: {"calls", "total_us", "counters"}}, "facts": ...}; FACTS.json is
what tools/prof_msm.py wrote about the run: sizes, MSM count, window) - the input of tools/pmc_accumulate.py."""
import collections
import json
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
if "--json" in sys.argv:
    i = sys.argv.index("--json")
    facts = json.load(open(sys.argv[i + 2])) if len(sys.argv) > i + 2 else None
    json.dump({"kernels": {k: {"calls": len(calls[k]), "total_us": round(dur[k], 2), "counters": dict(acc[k])} for k in acc}, "facts": facts},
              open(sys.argv[i + 1], "w"), indent=1, sort_keys=True)
