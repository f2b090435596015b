#!/usr/bin/env python3
"""Writes profiles/rNN_pmc_msm_accumulate.json FROM the round's raw PMC passes, so the summary cannot go stale against them
(the round-3 summary kept a "no kernel change" note after the kernel had changed).
usage: python tools/pmc_accumulate.py DIR TAG OUT.json
reads  DIR/TAG_pmc_<config>_<COUNTER[_COUNTER...]>.json  (tools/pmc_summary.py --json; <config> e.g. bn254_2p17, bls12381_2p14)
For msm_accumulate_kernel of every config: launches, MSMs and pairs of the run (tools/prof_msm.py's facts), the raw counter sums,
HBM-side bytes per pair = (2 x FETCH_SIZE + WRITE_SIZE) KB / pairs (MI355X_MICROARCH.md: FETCH_SIZE is doubled on gfx950),
VALU wave-instructions per MSM, and the wait share of the wave cycles."""
import glob
import json
import os
import re
import sys

d, tag, out = sys.argv[1], sys.argv[2], sys.argv[3]
configs = {}
for p in sorted(glob.glob(os.path.join(d, tag + "_pmc_*.json"))):
    m = re.match(re.escape(tag) + r"_pmc_((?:bn254|bls12381)_2p\d+(?:_c\d+)?)_([A-Z0-9_]+)\.json$", os.path.basename(p))
    if not m:
        continue
    t = json.load(open(p))
    acc = [(k, v) for k, v in t["kernels"].items() if k.startswith("msm_accumulate_kernel")]
    if not acc:
        continue
    name, row = acc[0]
    c = configs.setdefault(m.group(1), {"kernel": name, "raw": {}, "passes": []})
    c["passes"].append(os.path.basename(p))
    c["launches"], c["total_us"] = row["calls"], row["total_us"]          # the last pass read wins: they agree to a few percent
    for cn, v in row["counters"].items():
        c["raw"][cn + ("_KB" if cn in ("FETCH_SIZE", "WRITE_SIZE") else "")] = v
    if t.get("facts"):
        c["facts"] = t["facts"]
for name, c in configs.items():
    f = c.get("facts") or {}
    msms, pairs = f.get("msms_total"), f.get("pairs_total")
    c["msms"], c["pairs"], c["window_bits"] = msms, pairs, f.get("window_bits")
    c["algorithmic_bytes_per_pair"] = 96 if name.startswith("bn254") else 128
    r = c["raw"]
    if pairs and "FETCH_SIZE_KB" in r and "WRITE_SIZE_KB" in r:
        c["hbm_bytes_per_pair"] = round((2.0 * r["FETCH_SIZE_KB"] + r["WRITE_SIZE_KB"]) * 1024.0 / pairs, 1)
        c["table_bytes"] = f.get("table_bytes")
    if msms and "SQ_INSTS_VALU" in r:
        c["valu_instructions_per_msm"] = int(r["SQ_INSTS_VALU"] / msms)
        if pairs and f.get("windows"):
            c["valu_instructions_per_bucket_addition"] = round(r["SQ_INSTS_VALU"] * 64.0 / (pairs * f["windows"]), 1)
    if r.get("SQ_WAVE_CYCLES") and "SQ_WAIT_INST_ANY" in r:
        c["wait_inst_any_over_wave_cycles"] = round(r["SQ_WAIT_INST_ANY"] / r["SQ_WAVE_CYCLES"], 3)
res = dict(configs)
res["source"] = ("tools/pmc_accumulate.py over %s_pmc_*.json = tools/pmc_summary.py --json of `rocprofv3 --kernel-trace --pmc <one counter group per "
                 "pass> -- python tools/prof_msm.py ...` (tools/profile_round.sh); facts (MSMs, pairs, window) written by prof_msm.py itself" % tag)
res["correction"] = ("MI355X_MICROARCH.md HBM section: FETCH_SIZE tallies 64 B per request and is doubled on gfx950 (tools/ubench/gather64.hip: one "
                     "TCC_EA0_RDREQ and 64 B of FETCH_SIZE per gathered record whatever its size - the memory side moves 128 bytes per gather)")
json.dump(res, open(out, "w"), indent=1, sort_keys=True)
print(json.dumps({k: {x: v.get(x) for x in ("launches", "msms", "pairs", "window_bits", "hbm_bytes_per_pair", "valu_instructions_per_msm")} for k, v in configs.items()}, indent=1))
