#!/usr/bin/env python3
"""One G1 MSM at a time, 2^lo .. 2^hi (scalar, point) pairs, scalars resident in HBM: the measured single-GPU times behind
DESIGN.md section 6's projection for BASELINE.json configs[3] (one MSM sharded by index range over G GPUs: every rank runs an
MSM of n / G pairs, then ONE all-gather of a point per rank) - SURVEY.md section 8d config 4 "also report 2^20-2^24 to show
where sharding pays".
usage: python tools/msm_size_sweep.py [lo=11] [hi=24] [bn254|bls12_381] [out.json] [windows: e.g. 0,16,18,19,20 - 0 = library default]
Inputs: 2^16 distinct SRS-shaped bases [tau^i]G1 repeated to the size (the time of an MSM does not depend on the base values;
a repeated base meets its twin in a bucket with probability 2^-15 per pair and is handled by the ordinary addition), scalars =
random 253-bit integers (below r on both curves).  Each size: an MSM-only context (apk_msm_ctx_create: windowed tables resident),
3 warm-up MSMs, median and minimum of 9."""
import ctypes as C
import json
import os
import statistics
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from algoplonk_amd import ecc, setup, workloads
from algoplonk_amd._lib import lib, check

lo = int(sys.argv[1]) if len(sys.argv) > 1 else 11
hi = int(sys.argv[2]) if len(sys.argv) > 2 else 24
cv = ecc.BLS12_381 if len(sys.argv) > 3 and sys.argv[3] == "bls12_381" else ecc.BN254
out_path = sys.argv[4] if len(sys.argv) > 4 and sys.argv[4] != "-" else None
windows = [int(x) for x in sys.argv[5].split(",")] if len(sys.argv) > 5 else [0]
nb = 2 * cv.fp_bytes
DISTINCT = 1 << 16
tau = workloads.tau_from_seed(0xA192, cv.r)
srs = setup.unsafe_srs(cv, DISTINCT - 3, tau)
block = srs.g1[: DISTINCT * nb]
rows = []
for log_n in range(lo, hi + 1):
    n = 1 << log_n
    bases = block[: n * nb] if n <= DISTINCT else block * (n // DISTINCT)
    raw = bytearray(os.urandom(n * 32))
    raw[31::32] = bytes(b & 0x1F for b in raw[31::32])          # little-endian limbs: clear the top 3 bits -> < 2^253 < r
    d = None
    row = {"log_n": log_n}
    for c in windows:
        ctx = C.c_void_p()
        t0 = time.perf_counter()
        rc = lib.apk_msm_ctx_create(cv.abi, 0, bases, n, c, C.byref(ctx))
        if rc != 0:                                  # e.g. a window whose two-level sort layout does not fit this size
            row["c%d_error" % c] = lib.apk_last_error().decode()[:120]
            continue
        d = C.c_void_p()
        check(lib.apk_device_alloc(ctx, n * 32, C.byref(d)))
        check(lib.apk_device_upload(ctx, d, bytes(raw), n * 32))
        out = C.create_string_buffer(nb)
        for _ in range(3):
            check(lib.apk_msm_g1_device(ctx, 0, d, n, out))
        setup_s = time.perf_counter() - t0
        ts = []
        for _ in range(9):
            t1 = time.perf_counter()
            check(lib.apk_msm_g1_device(ctx, 0, d, n, out))
            ts.append((time.perf_counter() - t1) * 1e3)
        check(lib.apk_device_free(ctx, d))
        lib.apk_ctx_destroy(ctx)
        key = "" if c == 0 else "c%d_" % c
        row.update({key + "ms_median": round(statistics.median(ts), 4), key + "ms_min": round(min(ts), 4),
                    key + "mscalar_per_s": round(n / statistics.median(ts) / 1e3, 2), key + "setup_s": round(setup_s, 2),
                    key + "result_sha": __import__("hashlib").sha256(out.raw).hexdigest()[:12]})
    if len({v for k, v in row.items() if k.endswith("result_sha")}) > 1:
        raise SystemExit("window sizes disagree on the result at 2^%d: %r" % (log_n, row))
    rows.append(row)
    print(json.dumps(row), flush=True)
    del bases, raw
res = {"curve": cv.name, "what": "one apk_msm_g1_device at a time on an MSM-only context (window = library default), scalars resident; host wall clock per call",
       "rows": rows}
# the sharded MSM of configs[3] on G GPUs: every rank an MSM of n / G pairs + one exchange of a point per rank
by = {r["log_n"]: r["ms_median"] for r in rows if "ms_median" in r}
proj = []
for log_n in range(max(lo + 3, 14), hi + 1):
    if log_n not in by:
        continue
    line = {"log_n": log_n, "T1_ms": by[log_n]}
    for g, lg in ((2, 1), (4, 2), (8, 3)):
        if log_n - lg in by:
            line["T%d_msm_ms" % g] = by[log_n - lg]
            line["speedup_x%d_before_exchange" % g] = round(by[log_n] / by[log_n - lg], 2)
    proj.append(line)
res["sharded_projection"] = proj
if out_path:
    json.dump(res, open(out_path, "w"), indent=1)
print(json.dumps(res["sharded_projection"]))
