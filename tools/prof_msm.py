#!/usr/bin/env python3
"""Small driver for profiler runs: one context at BN254 2^log_n, then a few single MSMs and proofs.
usage: python tools/prof_msm.py [log_n] [msms] [proofs] [bn254|bls12_381]
APK_PROF_SLOTS=k: proving slots of the context (1).  APK_PROF_STATS=1: the proofs run with the library's HIP-event statistics on (as
bench.py's roofline pass does) and the average msm_accumulate_kernel launch time by those events is printed - under rocprofv3 the same
run then gives the profiler's average for the same launches (profiles/rNN_kernel_trace_*_lone_proofs.txt).  APK_PROF_FACTS=file: what the run did (sizes, MSMs, pairs, window, table bytes) as JSON, for tools/pmc_summary.py --json."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from algoplonk_amd import _lib, ecc, frontend, plonk, setup, workloads
from algoplonk_amd._lib import lib, check

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 17
n_msm = int(sys.argv[2]) if len(sys.argv) > 2 else 5
n_proofs = int(sys.argv[3]) if len(sys.argv) > 3 else 2
cv = ecc.BLS12_381 if len(sys.argv) > 4 and sys.argv[4] == "bls12_381" else ecc.BN254
wl = workloads.random_circuit(cv, log_n, 0xA190 if cv is ecc.BN254 else 0xA191)
n = wl.ccs.domain_size()
srs = setup.unsafe_srs(cv, n, wl.tau)
pk, vk = plonk.Setup(wl.ccs, srs, slots=int(os.environ.get("APK_PROF_SLOTS", "1")))
L, R, O = frontend.wire_columns(wl.ccs, wl.solution)
d = []
for v in (L, R, O):
    b = cv.fr_vector(v)
    p = C.c_void_p()
    check(lib.apk_device_alloc(pk.ctx, len(b), C.byref(p)))
    check(lib.apk_device_upload(pk.ctx, p, b, len(b)))
    d.append(p)
out = C.create_string_buffer(2 * cv.fp_bytes)
for _ in range(n_msm):
    check(lib.apk_msm_g1_device(pk.ctx, 0, d[0], n, out))
pr = _lib.Proof()
if os.environ.get("APK_PROF_STATS") == "1":
    pk.enable_stats(True)
    pk.stats(reset=True)
for _ in range(n_proofs):
    check(lib.apk_prove_device(pk.ctx, d[0], d[1], d[2], cv.fr_vector(wl.witness.public), cv.fr_vector(wl.blinding), None, C.byref(pr)))
if os.environ.get("APK_PROF_STATS") == "1":
    st = pk.stats(reset=True)
    print("HIP events: %d msm_accumulate_kernel launches of the proofs, avg %.4f ms, %.1f pairs per launch" %
          (st.msm_accumulate_launches, st.msm_accumulate_ms / max(st.msm_accumulate_launches, 1), st.msm_pairs / max(st.msm_accumulate_launches, 1)))
if os.environ.get("APK_PROF_FACTS"):
    c = pk.msm_window                                    # what the context chose (or APK_MSM_WINDOW)
    windows = (cv.r.bit_length() + 1 + c - 1) // c
    msms = 8 + n_msm + 9 * n_proofs           # Setup's 8 VK commitments + the single MSMs + 9 commitments per proof
    json.dump({"curve": cv.name, "log_n": log_n, "n": n, "single_msms": n_msm, "proofs": n_proofs, "msms_total": msms,
               "pairs_total": msms * (n + 2), "window_bits": c, "windows": windows, "table_bytes": (n + 3) * windows * 2 * cv.fp_bytes},
              open(os.environ["APK_PROF_FACTS"], "w"))
print("done")
