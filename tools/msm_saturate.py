#!/usr/bin/env python3
"""How many single 2^log_n MSMs per second the device sustains with T host threads (one context, T slots):
tells how much of a saturated proof's time the commitments account for.
usage: python tools/msm_saturate.py [log_n] [threads,threads,...] [msms per thread]"""
import ctypes as C
import os
import sys
import threading
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from algoplonk_amd import _lib, ecc, frontend, plonk, setup, workloads
from algoplonk_amd._lib import lib, check

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 17
tlist = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,8,24").split(",")]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
cv = ecc.BN254
wl = workloads.random_circuit(cv, log_n, 0xA190)
n = wl.ccs.domain_size()
srs = setup.unsafe_srs(cv, n, wl.tau)
pk, vk = plonk.Setup(wl.ccs, srs, slots=max(tlist))
L, R, O = frontend.wire_columns(wl.ccs, wl.solution)
b = cv.fr_vector(L)
p = C.c_void_p()
check(lib.apk_device_alloc(pk.ctx, len(b), C.byref(p)))
check(lib.apk_device_upload(pk.ctx, p, b, len(b)))


def worker(k):
    out = C.create_string_buffer(64)
    for _ in range(k):
        check(lib.apk_msm_g1_device(pk.ctx, 0, p, n, out))


worker(3)
for T in tlist:
    th = [threading.Thread(target=worker, args=(reps,)) for _ in range(T)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    print("threads %2d: %8.1f MSM/s  (%.3f ms per MSM, device-wide)" % (T, T * reps / dt, 1e3 * dt / (T * reps)), flush=True)
