#!/usr/bin/env python3
"""What each MSM phase costs AT SATURATION: the single-MSM rate with T host threads (one 2^log_n MSM each at a time) with one
phase - or several - not launched.  Needs a -DAPK_DEBUG_KNOCKOUT build (tools/build_variant.sh knockout "-DAPK_DEBUG_KNOCKOUT",
APK_LIB=.../libapk_knockout.so); the skipped phases leave garbage or the previous batch's data behind, which is the point: the
rate with a phase missing, against the full rate, is that phase's marginal cost with everything else still competing for the GPU.
usage: APK_LIB=$PWD/algoplonk_amd/libapk_knockout.so python tools/knockout.py [log_n] [threads] [msms per thread]"""
import ctypes as C
import os
import sys
import threading
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from algoplonk_amd import ecc, frontend, plonk, setup, workloads
from algoplonk_amd._lib import lib, check

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 17
T = int(sys.argv[2]) if len(sys.argv) > 2 else 16
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
cv = ecc.BN254
wl = workloads.random_circuit(cv, log_n, 0xA190)
n = wl.ccs.domain_size()
srs = setup.unsafe_srs(cv, n, wl.tau)
pk, vk = plonk.Setup(wl.ccs, srs, slots=T)
L, R, O = frontend.wire_columns(wl.ccs, wl.solution)
b = cv.fr_vector(L)
p = C.c_void_p()
check(lib.apk_device_alloc(pk.ctx, len(b), C.byref(p)))
check(lib.apk_device_upload(pk.ctx, p, b, len(b)))


def worker(k):
    out = C.create_string_buffer(64)
    for _ in range(k):
        check(lib.apk_msm_g1_device(pk.ctx, 0, p, n, out))


def rate():
    th = [threading.Thread(target=worker, args=(reps,)) for _ in range(T)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    return (time.perf_counter() - t0) / (T * reps) * 1e3


os.environ["APK_DEBUG_SKIP"] = "0"
rate()                                       # every slot has sorted a batch: the skipped sort phases leave valid data behind
NAMES = {1: "count pass + column scan (two-level: the whole sort)", 32: "scan launches", 64: "scatter pass", 97: "whole sort", 2: "accumulate", 4: "merge",
         8: "row/column sums", 16: "bit sums + final", 28: "whole tail", 125: "everything but accumulate", 127: "everything"}
base = None
for mask in (0, 1, 32, 64, 97, 2, 4, 8, 16, 28, 125, 127, 0):
    os.environ["APK_DEBUG_SKIP"] = str(mask)
    ms = rate()
    if base is None:
        base = ms
    print("skip %3d %-28s %.4f ms per MSM device-wide  (marginal %.4f ms = %4.1f %%)" % (mask, NAMES.get(mask, "-"), ms, base - ms, 100 * (base - ms) / base), flush=True)
