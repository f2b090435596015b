"""cpu_baseline leg of bench.py: time the oracle's plain-C prover (oracle/apk_oracle.c, kind = "port") on this box's
host cores, on the same workload the GPU just proved.  This is the ONLY place outside tests/ and smoke() that loads
anything under oracle/, and only as the reported baseline - never as part of the measured or shipped path."""
from __future__ import annotations

import os
import time


def cpu_baseline_prove(wl, srs, budget_s: float = 20.0, threads: int = 0):
    from algoplonk_amd import frontend
    from oracle import c_oracle

    lib = c_oracle.load()
    cv = wl.curve
    cores = threads or (os.cpu_count() or 1)
    tr = frontend.build_trace(wl.ccs)
    L, R, O = frontend.wire_columns(wl.ccs, wl.solution)
    cols = [cv.fr_vector(x) for x in (tr.ql, tr.qr, tr.qm, tr.qo, tr.qk)]
    args = (lib, cv.abi, tr.n, wl.ccs.GetNbPublicVariables(), srs.g1, cols, tr.perm, cv.fr_vector(L), cv.fr_vector(R),
            cv.fr_vector(O), cv.fr_vector(wl.witness.public), cv.fr_vector(wl.blinding))
    done, t0 = 0, time.perf_counter()
    blob = b""
    while True:
        rc, blob, _ = c_oracle.prove(*args, threads=cores)
        if rc != 0:
            raise RuntimeError("C oracle prover returned %d" % rc)
        done += 1
        el = time.perf_counter() - t0
        if el >= budget_s or el + el / done > 1.5 * budget_s:
            break
    return {"value": round(done / el, 5), "unit": "proofs/sec", "cores": cores, "kind": "port",
            "sample": "%d proof(s) of the same %s in %.1f s, oracle/apk_oracle.c with %d pthreads" % (done, wl.name, el, cores),
            "proof_sha256_prefix": __import__("hashlib").sha256(blob).hexdigest()[:16]}


def cpu_baseline_msm(cv, bases: bytes, scalars: bytes, n: int, budget_s: float = 20.0, threads: int = 0):
    """MSM-only baseline for bench.py --mode msm-sharded: the C oracle's Pippenger (orc_msm) over the same
    points and scalars on this box's host cores."""
    import ctypes as C
    from oracle import c_oracle

    lib = c_oracle.load()
    cores = threads or (os.cpu_count() or 1)
    out = C.create_string_buffer(2 * cv.fp_bytes)
    done, t0 = 0, time.perf_counter()
    while True:
        rc = lib.orc_msm(cv.abi, bases, scalars, n, cores, out)
        if rc != 0:
            raise RuntimeError("C oracle MSM returned %d" % rc)
        done += 1
        el = time.perf_counter() - t0
        if el >= budget_s or el + el / done > 1.5 * budget_s:
            break
    return {"value": round(done * n / el / 1e6, 4), "unit": "Mscalar/s", "cores": cores, "kind": "port",
            "sample": "%d MSM(s) of 2^%d pairs in %.1f s, oracle/apk_oracle.c Pippenger with %d pthreads" % (done, n.bit_length() - 1, el, cores),
            "result": out.raw}
