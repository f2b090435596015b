#!/usr/bin/env python3
"""Latency of ONE proof at a time on a context of several slots (the forms a lone proof gets on a serving context): median and
minimum over `reps` proofs, and the SHA-256 of the proof.  The run-time knobs are read once per process, so A/B runs are separate
processes:   APK_HOST_GLV=0 python tools/lone_latency.py 17 bn254
usage: python tools/lone_latency.py [log_n] [bn254|bls12_381] [reps] [slots]"""
import ctypes as C
import hashlib
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from algoplonk_amd import _lib, ecc, frontend, plonk, setup, workloads
from algoplonk_amd._lib import lib, check

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 17
cv = ecc.BLS12_381 if len(sys.argv) > 2 and sys.argv[2] == "bls12_381" else ecc.BN254
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
slots = int(sys.argv[4]) if len(sys.argv) > 4 else 16
wl = workloads.random_circuit(cv, log_n, 0xA190 if cv is ecc.BN254 else 0xA191)
srs = setup.unsafe_srs(cv, wl.ccs.domain_size(), wl.tau)
pk, vk = plonk.Setup(wl.ccs, srs, slots=slots)
L, R, O = frontend.wire_columns(wl.ccs, wl.solution)
d = []
for v in (L, R, O):
    b = cv.fr_vector(v)
    p = C.c_void_p()
    check(lib.apk_device_alloc(pk.ctx, len(b), C.byref(p)))
    check(lib.apk_device_upload(pk.ctx, p, b, len(b)))
    d.append(p)
pub, bl = cv.fr_vector(wl.witness.public), cv.fr_vector(wl.blinding)
pr = _lib.Proof()
for _ in range(5):
    check(lib.apk_prove_device(pk.ctx, d[0], d[1], d[2], pub, bl, None, C.byref(pr)))
pk.paths(reset=True)
ts = []
for _ in range(reps):
    t0 = time.perf_counter()
    check(lib.apk_prove_device(pk.ctx, d[0], d[1], d[2], pub, bl, None, C.byref(pr)))
    ts.append((time.perf_counter() - t0) * 1e3)
# the same proofs with the library's host-side timers on (they add synchronisations: read the shares, not the total)
pk.enable_stats(True)
pk.stats(reset=True)
for _ in range(10):
    check(lib.apk_prove_device(pk.ctx, d[0], d[1], d[2], pub, bl, None, C.byref(pr)))
st = pk.stats()
pk.enable_stats(False)
timers = {"prove_ms": round(st.prove_ms / st.proofs, 4), "round_ms": [round(x / st.proofs, 4) for x in st.round_ms],
          "host_lincomb_ms": round(st.host_lincomb_ms / st.proofs, 4)}
knobs = {k: v for k, v in os.environ.items() if k.startswith("APK_")}
print(json.dumps({"curve": cv.name, "log_n": log_n, "slots": slots, "reps": reps, "median_ms": round(statistics.median(ts), 4), "min_ms": round(min(ts), 4),
                  "sha256": hashlib.sha256(bytes(pr)).hexdigest()[:16], "pooled": pk.paths()["host_lincomb_pooled"], "instrumented": timers, "knobs": knobs}))
