#!/usr/bin/env python3
"""Soak: T threads prove the same BN254 2^log_n instance over and over on one context; every proof must have the same
bytes (same inputs + same blinding scalars -> byte-identical proofs), so any race between slots shows up as a second hash.
usage: python tools/soak.py [log_n] [threads] [rounds] [bn254|bls12_381] [slots]
(slots defaults to the thread count; fewer slots than threads also exercises the wait at the slot gate, two or three threads the
host-side paths of a nearly idle context: parked threads, the early [H] part)"""
import ctypes as C
import hashlib
import os
import sys
import threading

os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from algoplonk_amd import _lib, ecc, frontend, plonk, setup, workloads
from algoplonk_amd._lib import lib, check

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 17
T = int(sys.argv[2]) if len(sys.argv) > 2 else 32
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 20
cv = ecc.BLS12_381 if len(sys.argv) > 4 and sys.argv[4] == "bls12_381" else ecc.BN254
slots = int(sys.argv[5]) if len(sys.argv) > 5 else T
wl = workloads.random_circuit(cv, log_n, 0xA190 if cv is ecc.BN254 else 0xA191)
n = wl.ccs.domain_size()
srs = setup.unsafe_srs(cv, n, wl.tau)
pk, vk = plonk.Setup(wl.ccs, srs, slots=slots)
L, R, O = frontend.wire_columns(wl.ccs, wl.solution)
dptr = []
for b in (cv.fr_vector(v) for v in (L, R, O)):
    p = C.c_void_p()
    check(lib.apk_device_alloc(pk.ctx, len(b), C.byref(p)))
    check(lib.apk_device_upload(pk.ctx, p, b, len(b)))
    dptr.append(p)
pub = cv.fr_vector(wl.witness.public)
bl = cv.fr_vector(wl.blinding)
hashes = {}
lock = threading.Lock()


def worker():
    pr = _lib.Proof()
    out = C.create_string_buffer(2048)
    ln = C.c_size_t(0)
    for _ in range(rounds):
        check(lib.apk_prove_device(pk.ctx, dptr[0], dptr[1], dptr[2], pub, bl, None, C.byref(pr)))
        check(lib.apk_marshal_proof(C.byref(pr), out, 2048, C.byref(ln)))
        h = hashlib.sha256(out.raw[: ln.value]).hexdigest()[:16]
        with lock:
            hashes[h] = hashes.get(h, 0) + 1


th = [threading.Thread(target=worker) for _ in range(T)]
for t in th: t.start()
for t in th: t.join()
print("proofs %d, distinct hashes %d: %s" % (sum(hashes.values()), len(hashes), hashes))
sys.exit(0 if len(hashes) == 1 else 1)
