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
    # A single proof spread over every core is the WORST way to use a many-core host (coarse pthread tasks, serial sections): run P
    # proofs concurrently with cores / P threads each, for a few P, and report the best aggregate rate - the CPU analogue of the
    # GPU leg's concurrent callers.
    import hashlib
    import threading
    rc, blob, _ = c_oracle.prove(*args, threads=cores)           # also initialises the library's tables before any concurrency
    if rc != 0:
        raise RuntimeError("C oracle prover returned %d" % rc)
    sha = hashlib.sha256(blob).hexdigest()[:16]
    plans = sorted({p for p in (1, 8) if p <= max(1, cores // 2)} | {1})   # 32 x 8 threads: slower than 8 x 32 and a minute per try
    best, tried = None, []
    slice_s = budget_s / len(plans)
    for P in plans:
        per = max(1, cores // P)
        counts = [0] * P
        stop = time.perf_counter() + slice_s
        bad = []

        def worker(i, per=per, stop=stop):
            while True:
                rc_, _, _ = c_oracle.prove(*args, threads=per)
                if rc_ != 0:
                    bad.append(rc_)
                    return
                counts[i] += 1
                if time.perf_counter() >= stop:
                    return

        t0 = time.perf_counter()
        ts = [threading.Thread(target=worker, args=(i,)) for i in range(P)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        el = time.perf_counter() - t0
        if bad:
            raise RuntimeError("C oracle prover returned %d" % bad[0])
        rate = sum(counts) / el
        tried.append({"concurrent_proofs": P, "threads_each": per, "proofs": sum(counts), "seconds": round(el, 1), "proofs_per_sec": round(rate, 4)})
        if best is None or rate > best[0]:
            best = (rate, P, per, sum(counts), el)
    rate, P, per, done, el = best
    return {"value": round(rate, 5), "unit": "proofs/sec", "cores": cores, "kind": "port",
            "sample": "%d proof(s) of the same %s in %.1f s, oracle/apk_oracle.c: %d concurrent proofs x %d pthreads (best of %s)"
                      % (done, wl.name, el, P, per, [t["concurrent_proofs"] for t in tried]),
            "tried": tried, "proof_sha256_prefix": sha}


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
