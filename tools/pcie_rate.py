#!/usr/bin/env python3
"""Proof latency / throughput with HOST-resident L,R,O (apk_prove: 3 x n x 32 B over PCIe per proof) next to the
device-resident variant bench.py times (apk_prove_device).  DESIGN.md §7 quotes the result; it is never bench `value`."""
import ctypes as C, os, sys, threading, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from algoplonk_amd import _lib, ecc, frontend, plonk, setup, workloads
from algoplonk_amd._lib import lib, check

cv = ecc.BN254
wl = workloads.random_circuit(cv, 17, 0xA190)
n = wl.ccs.domain_size()
INF = 24
pk, vk = plonk.Setup(wl.ccs, setup.unsafe_srs(cv, n, wl.tau), slots=INF)
L, R, O = (cv.fr_vector(v) for v in frontend.wire_columns(wl.ccs, wl.solution))
pub, bl = cv.fr_vector(wl.witness.public), cv.fr_vector(wl.blinding)
d = []
for b in (L, R, O):
    p = C.c_void_p(); check(lib.apk_device_alloc(pk.ctx, len(b), C.byref(p))); check(lib.apk_device_upload(pk.ctx, p, b, len(b))); d.append(p)
outs = [_lib.Proof() for _ in range(INF)]
def host(i): check(lib.apk_prove(pk.ctx, L, R, O, pub, bl, None, C.byref(outs[i])))
def dev(i): check(lib.apk_prove_device(pk.ctx, d[0], d[1], d[2], pub, bl, None, C.byref(outs[i])))
for name, fn in (("device-resident", dev), ("host buffers (PCIe)", host)):
    fn(0); t = time.perf_counter()
    for _ in range(10): fn(0)
    lat = (time.perf_counter() - t) / 10
    def step():
        ts = [threading.Thread(target=fn, args=(i,)) for i in range(INF)]
        [x.start() for x in ts]; [x.join() for x in ts]
    step(); t = time.perf_counter()
    for _ in range(4): step()
    thr = 4 * INF / (time.perf_counter() - t)
    print("%-22s latency %.3f ms   throughput %.1f proofs/s (%d in flight)" % (name, lat * 1e3, thr, INF))
