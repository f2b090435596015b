"""cpu_baseline leg of bench.py: time the oracle's host provers (oracle/apk_oracle.c clarity-first, oracle/fast_prover.c
performance-first; kind = "port") on this box's host cores, on the same workload the GPU just proved.  This is the ONLY place outside tests/ and smoke() that loads
anything under oracle/, and only as the reported baseline - never as part of the measured or shipped path."""
from __future__ import annotations

import os
import time


def effective_cores():
    """CPUs this process can actually use: the smaller of the visible CPUs, the affinity mask and the cgroup CPU quota.  The GPU
    boxes of this build show 256 CPUs (2 x EPYC 9575F) behind a quota of 16 (`cpu.max` = 1600000 100000): 256 threads there are
    sixteen cores' worth of time slices, and rounds 1-4 reported `cores: 256` for what 16 could do."""
    n = os.cpu_count() or 1
    info = {"visible_cpus": n}
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            quota, period = open(path).read().split()[:2]
            info["cgroup_cpu_max"] = "%s %s" % (quota, period)
            if quota != "max":
                n = min(n, max(1, int(int(quota) / int(period))))
        except Exception:
            pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            info["cgroup_cfs_quota"] = "%d %d" % (q, per)
            n = min(n, max(1, q // per))
    except Exception:
        pass
    return n, info


def cpu_baseline_prove(wl, srs, budget_s: float = 20.0, threads: int = 0, check_against_plain: bool = True):
    return cpu_baseline_prove_ccs(wl.curve, wl.name, wl.ccs, srs, wl.solution, wl.witness.public, wl.blinding, (), budget_s, threads,
                                  check_against_plain)


def cpu_baseline_prove_items(cv, name, ccs, srs, item, budget_s: float = 20.0, threads: int = 0, check_against_plain: bool = True):
    """The leg on an already packed assignment (algoplonk_amd.batch.Assignment: the very bytes the GPU proved)."""
    return cpu_baseline_prove_ccs(cv, name, ccs, srs, None, None, None, (), budget_s, threads, check_against_plain, packed=item)


def cpu_baseline_prove_ccs(cv, name, ccs, srs, solution, public, blinding, pi2_cols=(), budget_s: float = 20.0, threads: int = 0,
                           check_against_plain: bool = True, packed=None):
    """The same leg for any constraint system, BSB22 commitments included (`pi2_cols` = the committed columns the solver made:
    the performance-first prover recomputes their commitments over the Lagrange SRS and must arrive at the GPU's bytes; the
    clarity-first C oracle has no BSB22 path, so it is skipped there)."""
    from algoplonk_amd import frontend
    from oracle import c_oracle

    lib = c_oracle.load()
    eff, cpu_info = effective_cores()
    cores = threads or eff
    tr = frontend.build_trace(ccs)
    cols = [cv.fr_vector(x) for x in (tr.ql, tr.qr, tr.qm, tr.qo, tr.qk)]
    nbc = len(ccs.commitments)
    if nbc:
        check_against_plain = False
        if not srs.g1_lagrange:
            raise RuntimeError("BSB22 circuit without a Lagrange SRS")
    if packed is not None:
        args = (lib, cv.abi, tr.n, ccs.GetNbPublicVariables(), srs.g1, cols, tr.perm, packed.L, packed.R, packed.O, packed.public, packed.blinding)
        pi2_b = list(packed.pi2)
    else:
        L, R, O = frontend.wire_columns(ccs, solution)
        args = (lib, cv.abi, tr.n, ccs.GetNbPublicVariables(), srs.g1, cols, tr.perm, cv.fr_vector(L), cv.fr_vector(R),
                cv.fr_vector(O), cv.fr_vector(public), cv.fr_vector(blinding))
        pi2_b = [cv.fr_vector(p) for p in pi2_cols]
    # Two host provers, both ports (neither is gnark): the clarity-first orc_prove - the parity oracle, one proof for the hash and
    # as the lower bracket - and the performance-first orc_fast_prove (oracle/fast_prover.c: circuit-only work hoisted into a
    # context, batch-affine Pippenger, parallel FFTs, one-coset quotient), held to orc_prove's bytes by tests/test_oracle_c.py and
    # again right here.  A single proof spread over every core is the WORST way to use a many-core host, so P proofs run
    # concurrently with cores / P threads each, for a few P, and the best aggregate rate is reported - the CPU analogue of the GPU
    # leg's concurrent callers.
    import hashlib
    import threading
    tried, plain_s, blob = [], 0.0, None
    if check_against_plain:      # (skipped at the largest sizes, where one clarity-first proof alone takes minutes)
        t0 = time.perf_counter()
        rc, blob, _ = c_oracle.prove(*args, threads=cores)           # also initialises the library's tables before any concurrency
        plain_s = time.perf_counter() - t0
        if rc != 0:
            raise RuntimeError("C oracle prover returned %d" % rc)
        tried.append({"prover": "oracle/apk_oracle.c orc_prove (clarity-first)", "concurrent_proofs": 1, "threads_each": cores, "proofs": 1,
                      "seconds": round(plain_s, 2), "proofs_per_sec": round(1.0 / plain_s, 4)})
    t0 = time.perf_counter()
    fp = c_oracle.FastProver(lib, cv.abi, tr.n, ccs.GetNbPublicVariables(), srs.g1, cols, tr.perm, threads=cores,
                             qcp=[cv.fr_vector(q) for q in tr.qcp[:nbc]], cci=[cidx for _, cidx in ccs.commitments],
                             srs_lagrange=srs.g1_lagrange if nbc else None)
    setup_s = time.perf_counter() - t0
    pargs = args[7:]
    t0 = time.perf_counter()
    rc, fblob, _ = fp.prove(*pargs, threads=cores, pi2=pi2_b)
    first_s = time.perf_counter() - t0
    if rc != 0 or (blob is not None and fblob != blob):
        raise RuntimeError("the fast host prover disagrees with the oracle (rc %d)" % rc)
    blob = fblob
    sha = hashlib.sha256(blob).hexdigest()[:16]
    plans = [p for p in (1, 2, 4, 8) if p <= max(1, cores // 2)] or [1]
    if first_s * 3 > budget_s:       # a proof is a sizeable part of the budget: one plan only (one proof over all cores)
        plans = [1]
    best = None
    slice_s = max(2.0, (budget_s - plain_s - setup_s) / len(plans))
    for P in plans:
        per = max(1, cores // P)
        counts = [0] * P
        stop = time.perf_counter() + slice_s
        bad = []

        def worker(i, per=per, stop=stop):
            while True:
                rc_, b_, _ = fp.prove(*pargs, threads=per, pi2=pi2_b)
                if rc_ != 0 or b_ != blob:
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
            raise RuntimeError("the fast host prover returned %r or another proof" % bad[0])
        rate = sum(counts) / el
        tried.append({"prover": "oracle/fast_prover.c orc_fast_prove (performance-first)", "concurrent_proofs": P, "threads_each": per,
                      "proofs": sum(counts), "seconds": round(el, 1), "proofs_per_sec": round(rate, 4)})
        if best is None or rate > best[0]:
            best = (rate, P, per, sum(counts), el)
    fp.close()
    rate, P, per, done, el = best
    return {"value": round(rate, 5), "unit": "proofs/sec", "cores": cores, "kind": "port",
            "sample": "%d proof(s) of the same %s in %.1f s, oracle/fast_prover.c (batch-affine Pippenger, parallel FFTs, circuit-only work "
                      "hoisted: %.1f s once): %d concurrent proofs x %d pthreads on %d usable cores (best of %s); %s"
                      % (done, name, el, setup_s, P, per, cores, [t["concurrent_proofs"] for t in tried if "fast" in t["prover"]],
                         "every proof's bytes = the clarity-first oracle's" if check_against_plain else
                         "every proof's bytes = its first proof's (the clarity-first oracle was not run: size, or a BSB22 circuit it has no path for)"),
            "tried": tried, "proof_sha256_prefix": sha, "cpu": cpu_info}


def oracle_blobs(cv, ccs, srs, items, threads: int = 0, check_first_against_plain: bool = False):
    """The CHECKER of the distinct-witness runs (bench.py's timed region, tools/soak.py, the under-load parity tests): the host
    prover's proof blob for EVERY assignment in `items` (algoplonk_amd.batch.Assignment: L, R, O, public, blinding, pi2 bytes) of
    one circuit - oracle/fast_prover.c, whose bytes tests/test_oracle_c.py holds to the clarity-first oracle's (and which the
    first assignment is held to again here on request).  Circuit-only work once, then ~0.4 s per 2^17 proof on 16 cores."""
    from algoplonk_amd import frontend
    from oracle import c_oracle

    lib = c_oracle.load()
    cores = threads or effective_cores()[0]
    tr = frontend.build_trace(ccs)
    nbc = len(ccs.commitments)
    cols = [cv.fr_vector(x) for x in (tr.ql, tr.qr, tr.qm, tr.qo, tr.qk)]
    fp = c_oracle.FastProver(lib, cv.abi, tr.n, ccs.GetNbPublicVariables(), srs.g1, cols, tr.perm, threads=cores,
                             qcp=[cv.fr_vector(q) for q in tr.qcp[:nbc]], cci=[cidx for _, cidx in ccs.commitments],
                             srs_lagrange=srs.g1_lagrange if nbc else None)
    blobs = []
    try:
        for i, it in enumerate(items):
            rc, blob, _ = fp.prove(it.L, it.R, it.O, it.public, it.blinding, threads=cores, pi2=list(it.pi2))
            if rc != 0:
                blobs.append(None)          # the host prover refuses an unsatisfying witness too
                continue
            if i == 0 and check_first_against_plain and not nbc:
                rc2, plain, _ = c_oracle.prove(lib, cv.abi, tr.n, ccs.GetNbPublicVariables(), srs.g1, cols, tr.perm, it.L, it.R, it.O,
                                               it.public, it.blinding, threads=cores)
                if rc2 != 0 or plain != blob:
                    raise RuntimeError("the fast host prover disagrees with the clarity-first oracle")
            blobs.append(blob)
    finally:
        fp.close()
    return blobs


def cpu_baseline_msm(cv, bases: bytes, scalars: bytes, n: int, budget_s: float = 20.0, threads: int = 0):
    """MSM-only baseline for bench.py --mode msm-sharded: the C oracle's Pippenger (orc_msm) over the same
    points and scalars on this box's host cores."""
    import ctypes as C
    from oracle import c_oracle

    lib = c_oracle.load()
    eff, cpu_info = effective_cores()
    cores = threads or eff
    out, ref = C.create_string_buffer(2 * cv.fp_bytes), C.create_string_buffer(2 * cv.fp_bytes)
    # the clarity-first Pippenger once (the checker and the lower bracket), then the performance-first one (oracle/fast_msm_tmpl.h:
    # batch-affine, signed digits, thread pool) for the budget
    t0 = time.perf_counter()
    if lib.orc_msm(cv.abi, bases, scalars, n, cores, ref) != 0:
        raise RuntimeError("C oracle MSM failed")
    plain_s = time.perf_counter() - t0
    done, t0 = 0, time.perf_counter()
    while True:
        rc = lib.orc_msm_fast(cv.abi, bases, scalars, n, cores, out)
        if rc != 0 or out.raw != ref.raw:
            raise RuntimeError("the fast host MSM returned %d or another point than the oracle's" % rc)
        done += 1
        el = time.perf_counter() - t0
        if el >= budget_s or el + el / done > 1.5 * budget_s:
            break
    return {"value": round(done * n / el / 1e6, 4), "unit": "Mscalar/s", "cores": cores, "kind": "port",
            "sample": "%d MSM(s) of 2^%d pairs in %.1f s, oracle/fast_msm_tmpl.h (batch-affine Pippenger) with %d pthreads; the clarity-first "
                      "orc_msm took %.2f s for one and gives the same point" % (done, n.bit_length() - 1, el, cores, plain_s),
            "clarity_first_mscalar_per_s": round(n / plain_s / 1e6, 4), "cpu": cpu_info,
            "result": out.raw}
