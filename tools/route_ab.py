#!/usr/bin/env python3
"""[L][R][O] over the canonical SRS against over the Lagrange SRS (APK_WIRES_LAGRANGE = 0 / 1), on a uniform and on a bit-heavy
witness: lone-proof latency and proofs/s with 32 callers, each combination in a context of its own (the knob is read at context
creation).  usage: python tools/route_ab.py [bn254|bls12_381] [log_n] [steps]"""
import ctypes as C
import json
import os
import sys
import threading
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from algoplonk_amd import _lib, ecc, frontend, plonk, setup, workloads
from algoplonk_amd._lib import lib, check

cv = ecc.BLS12_381 if len(sys.argv) > 1 and sys.argv[1] == "bls12_381" else ecc.BN254
log_n = int(sys.argv[2]) if len(sys.argv) > 2 else 17
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
T = 32
rows = []
for kind in ("uniform", "bits"):
    wl = (workloads.skewed_circuit if kind == "bits" else workloads.random_circuit)(cv, log_n, 0xA190)
    n = wl.ccs.domain_size()
    srs = setup.unsafe_srs(cv, n, wl.tau)
    L, R, O = frontend.wire_columns(wl.ccs, wl.solution)
    for mode in ("0", "1"):
        os.environ["APK_WIRES_LAGRANGE"] = mode
        pk, vk = plonk.Setup(wl.ccs, srs, slots=T)
        d = []
        for v in (L, R, O):
            b = cv.fr_vector(v)
            p = C.c_void_p()
            check(lib.apk_device_alloc(pk.ctx, len(b), C.byref(p)))
            check(lib.apk_device_upload(pk.ctx, p, b, len(b)))
            d.append(p)
        pub, bl = cv.fr_vector(wl.witness.public), cv.fr_vector(wl.blinding)

        def prove(pr):
            check(lib.apk_prove_device(pk.ctx, d[0], d[1], d[2], pub, bl, None, C.byref(pr)))

        pr = _lib.Proof()
        for _ in range(3):
            prove(pr)
        t0 = time.perf_counter()
        for _ in range(10):
            prove(pr)
        lone_ms = (time.perf_counter() - t0) / 10 * 1e3
        blob = bytes(pr)[:600]

        def worker():
            q = _lib.Proof()
            for _ in range(steps):
                prove(q)

        for rep in range(2):          # first pass warms up
            th = [threading.Thread(target=worker) for _ in range(T)]
            t0 = time.perf_counter()
            [t.start() for t in th]
            [t.join() for t in th]
            rate = T * steps / (time.perf_counter() - t0)
        row = {"curve": cv.name, "log_n": log_n, "witness": kind, "route": "lagrange" if mode == "1" else "canonical", "lone_ms": round(lone_ms, 3),
               "proofs_per_s_32_callers": round(rate, 1), "lagrange_batches": pk.paths()["msm_lagrange_wires"], "blob_head": blob[:8].hex()}
        rows.append(row)
        print(json.dumps(row), flush=True)
        pk.close()
