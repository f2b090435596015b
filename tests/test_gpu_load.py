"""-m gpu: slot isolation.  Concurrent callers with DISTINCT inputs, every blob held to the C oracle's proof of ITS OWN inputs.

Rounds 1-5 proved the loaded path with identical inputs in every caller: a slot that read another slot's workspace, pinned
result buffer or staging set read the same values and passed (VERDICT r05 "What's weak" #1).  Here every caller has its own
assignment of the one compiled circuit - its own L, R, O, public inputs, blinding scalars, BSB22 columns and hiding pairs, as
every `(*CompiledCircuit).Verify` call has (/root/reference/algoplonk.go:79-98) - and walks through the set, so that over the
rounds every slot sees every assignment next to every other.  Any cross-slot aliasing now produces bytes that are nobody's proof.
The same through `apk_prove` with host pointers (page-locked and ordinary memory: the call the cgo shim makes,
INTEGRATION.md), with an unsatisfying witness among the good ones, and on a circuit with a BSB22 commitment.
"""
import ctypes as C
import hashlib
import threading

import pytest

from algoplonk_amd import _lib, batch, frontend, plonk as ap_plonk, setup as ap_setup, workloads
from algoplonk_amd import MarshalProof
from algoplonk_amd._lib import lib, check
from bench_cpu import oracle_blobs

from helpers import CURVES, oracle_threads

pytestmark = pytest.mark.gpu


def _marshal(pr) -> bytes:
    out = C.create_string_buffer(2048)
    ln = C.c_size_t(0)
    check(lib.apk_marshal_proof(C.byref(pr), out, 2048, C.byref(ln)))
    return out.raw[: ln.value]


def _run_callers(ws, callers, rounds, where, pick, expect_error=None):
    """`callers` threads x `rounds` proofs; caller i proves assignment pick(i, round).  Returns {(assignment, blob): count} and the
    error list; assignments in `expect_error` must fail with that code (and are not recorded)."""
    got, errors, lock = {}, [], threading.Lock()
    expect_error = expect_error or {}

    def worker(i):
        pr = _lib.Proof()
        for r in range(rounds):
            a = pick(i, r)
            rc = ws.prove(a, pr, where)
            if a in expect_error:
                if rc != expect_error[a]:
                    errors.append(("assignment %d: expected status %d, got %d" % (a, expect_error[a], rc), lib.apk_last_error()))
                    return
                continue
            if rc != 0:
                errors.append((rc, lib.apk_last_error()))
                return
            blob = _marshal(pr)
            with lock:
                got[(a, blob)] = got.get((a, blob), 0) + 1

    th = [threading.Thread(target=worker, args=(i,)) for i in range(callers)]
    [t.start() for t in th]
    [t.join() for t in th]
    return got, errors


def _assert_every_blob_is_its_own_oracle_proof(got, want, what):
    wrong = [(a, hashlib.sha256(b).hexdigest()[:12], c) for (a, b), c in got.items() if b != want[a]]
    assert not wrong, "%s: %d blob(s) differ from the C oracle's proof of the SAME inputs: %s" % (what, len(wrong), wrong[:6])


@pytest.mark.parametrize("cname,log_n", [("bn254", 17), ("bls12-381", 14)])
def test_distinct_witnesses_under_load_match_the_c_oracle(gpu, cname, log_n):
    cv, ov = CURVES[cname]
    seed = 0xA190 if cname == "bn254" else 0xA191
    wl = workloads.random_circuit(cv, log_n, seed)
    n = wl.ccs.domain_size()
    srs = ap_setup.unsafe_srs(cv, n, wl.tau, device=gpu)
    T, K, rounds = 32, 8, 4
    pk, vk = ap_plonk.Setup(wl.ccs, srs, device=gpu, slots=T)
    ws = batch.WitnessSet(pk, wl.ccs, workloads.variants(wl, K, seed)).to_device().to_pinned(gpu)
    want = oracle_blobs(cv, wl.ccs, srs, ws.items, threads=oracle_threads(), check_first_against_plain=True)
    assert len(set(want)) == K, "the assignments are meant to be distinct"
    pick = lambda i, r: (i + 3 * r) % K          # every caller walks the set; neighbours hold different assignments at any time
    for where, rr in (("device", rounds), ("pinned", rounds), ("pageable", 2)):
        pk.paths(reset=True)
        got, errors = _run_callers(ws, T, rr, where, pick)
        assert not errors, errors[0]
        assert sum(got.values()) == T * rr
        _assert_every_blob_is_its_own_oracle_proof(got, want, "%s inputs" % where)
        assert {a for a, _ in got} == set(range(K))
        loaded = pk.paths(reset=True)
        assert loaded["proofs"] == T * rr
        assert loaded["host_inputs"] == (0 if where == "device" else T * rr), loaded
        # the loaded forms of the kernels made these blobs (as in test_proofs_under_load_match_the_c_oracle)
        assert loaded["msm_lean_tail"] >= loaded["msm_batches"] // 2, loaded
        assert loaded["msm_sort_two_level"] >= loaded["msm_batches"] // 2, loaded
    # a lone proof of every assignment from host memory: the latency forms, the same bytes
    pr = _lib.Proof()
    for a in range(K):
        check(ws.prove(a, pr, "pinned"))
        assert _marshal(pr) == want[a], a
    ws.close()
    pk.close()


@pytest.mark.parametrize("cname,log_n", [("bn254", 15), ("bls12-381", 13)])
@pytest.mark.parametrize("where", ["device", "pinned"])
def test_an_unsatisfying_witness_among_good_ones(gpu, cname, log_n, where):
    """One caller keeps handing in a witness that violates a gate while 31 others prove good, distinct ones on the same context:
    it gets APK_ERR_WITNESS every time, the others the C oracle's bytes for their own inputs, and the context stays usable."""
    cv, ov = CURVES[cname]
    seed = 0xBAD0 + log_n
    wl = workloads.random_circuit(cv, log_n, seed)
    n = wl.ccs.domain_size()
    srs = ap_setup.unsafe_srs(cv, n, wl.tau, device=gpu)
    T, K, rounds = 32, 6, 4
    pk, vk = ap_plonk.Setup(wl.ccs, srs, device=gpu, slots=T)
    ws = batch.WitnessSet(pk, wl.ccs, workloads.variants(wl, K + 1, seed)).to_device().to_pinned(gpu)
    ws.corrupt(K)                                                    # the last assignment no longer satisfies the circuit
    want = oracle_blobs(cv, wl.ccs, srs, ws.items[:K], threads=oracle_threads())
    pick = lambda i, r: K if i == 5 else (i + r) % K
    got, errors = _run_callers(ws, T, rounds, where, pick, expect_error={K: _lib.APK_ERR_WITNESS})
    assert not errors, errors[0]
    assert sum(got.values()) == (T - 1) * rounds
    _assert_every_blob_is_its_own_oracle_proof(got, want, "good callers beside a failing one (%s)" % where)
    pr = _lib.Proof()
    assert ws.prove(K, pr, where) == _lib.APK_ERR_WITNESS and b"does not satisfy" in lib.apk_last_error()
    for a in range(K):                                               # ... and afterwards: lone proofs, the oracle's bytes
        check(ws.prove(a, pr, where))
        assert _marshal(pr) == want[a]
    ws.close()
    pk.close()


@pytest.mark.parametrize("cname,log_n", [("bn254", 16), ("bls12-381", 12)])
def test_distinct_bsb22_witnesses_under_load(gpu, cname, log_n):
    """The same on a circuit with one BSB22 commitment: every caller has its own committed column (its own witness values in the
    committed rows, its own hiding pair) on top of its own wires - compared with the host prover's BSB22 path per assignment."""
    cv, ov = CURVES[cname]
    seed = 0xA193
    ccs, w, bl, tau = workloads.random_circuit_bsb22(cv, log_n, seed, nb_commitments=1)
    n = ccs.domain_size()
    srs = ap_setup.unsafe_srs(cv, n, tau, device=gpu, lagrange=True)
    T, K, rounds = 16, 4, 3
    pk, vk = ap_plonk.Setup(ccs, srs, device=gpu, slots=T)
    vs = [workloads.Variant(w, bl, None, [(0xA193, 0x3910A)])] + workloads.variant_inputs(ccs, K - 1, seed)
    ws = batch.WitnessSet(pk, ccs, vs).to_device().to_pinned(gpu)
    assert len({it.pi2[0] for it in ws.items}) == K
    want = oracle_blobs(cv, ccs, srs, ws.items, threads=oracle_threads())
    assert len(set(want)) == K
    pick = lambda i, r: (i + r) % K
    for where in ("device", "pinned"):
        got, errors = _run_callers(ws, T, rounds, where, pick)
        assert not errors, errors[0]
        assert sum(got.values()) == T * rounds
        _assert_every_blob_is_its_own_oracle_proof(got, want, "BSB22, %s inputs" % where)
    ws.close()
    pk.close()


def test_host_memory_entry_points(gpu):
    """apk_host_alloc / apk_host_register: page-locked memory the HIP runtime accepts as such, argument checking, and a proof from a
    REGISTERED ordinary buffer (what the cgo shim does with a Go-allocated pool) equal to the proof from device-resident inputs."""
    cv, ov = CURVES["bn254"]
    wl = workloads.random_circuit(cv, 10, 0x4057)
    srs = ap_setup.unsafe_srs(cv, wl.ccs.domain_size(), wl.tau, device=gpu)
    pk, vk = ap_plonk.Setup(wl.ccs, srs, device=gpu, slots=2)
    ws = batch.WitnessSet(pk, wl.ccs, workloads.variants(wl, 2, 0x4057)).to_device()
    pr = _lib.Proof()
    check(ws.prove(1, pr, "device"))
    want = _marshal(pr)
    it = ws.items[1]
    pool = C.create_string_buffer(3 * len(it.L))                       # an ordinary allocation, registered once
    check(lib.apk_host_register(pool, len(pool)))
    base = C.addressof(pool)
    for j, b in enumerate((it.L, it.R, it.O)):
        C.memmove(base + j * len(b), b, len(b))
    check(lib.apk_prove(pk.ctx, base, base + len(it.L), base + 2 * len(it.L), it.public, it.blinding, None, C.byref(pr)))
    assert _marshal(pr) == want
    check(lib.apk_host_unregister(pool))
    p = C.c_void_p()
    assert lib.apk_host_alloc(gpu, 0, C.byref(p)) == _lib.APK_ERR_ARG
    assert lib.apk_host_alloc(99, 64, C.byref(p)) == _lib.APK_ERR_ARG
    check(lib.apk_host_alloc(gpu, 1 << 20, C.byref(p)))
    assert p.value
    check(lib.apk_host_free(p))
    check(lib.apk_host_free(None))
    ws.close()
    pk.close()


# ---- gangs (csrc/gang.h): two to four proofs per stream, their MSM / NTT batches in ONE launch sequence -----------------------------
@pytest.mark.parametrize("cname,log_n,gang", [("bn254", 17, 2), ("bn254", 15, 4), ("bls12-381", 14, 4), ("bls12-381", 14, 3)])
def test_ganged_proofs_match_the_c_oracle(gpu, cname, log_n, gang, monkeypatch):
    """With more callers than streams a context pairs its callers up (APK_GANG members per stream): the members run the unchanged
    prover in lockstep on one stream and meet at every commitment batch and transform batch, where ONE launch sequence carries all
    their operands.  Same kernels, same arithmetic: every blob must still be the C oracle's proof of its OWN inputs - here with
    distinct assignments in the members of a gang, so a member reading its neighbour's sums out of the shared result area, or a
    merged launch mixing operands up, cannot hide - and the counters must show that merged launches made them.  From device and
    from host memory; one caller keeps handing in an unsatisfying witness and LEAVES its gang in round 3 every time."""
    cv, ov = CURVES[cname]
    seed = 0x6A06 + log_n
    monkeypatch.setenv("APK_GANG", str(gang))
    wl = workloads.random_circuit(cv, log_n, seed)
    n = wl.ccs.domain_size()
    srs = ap_setup.unsafe_srs(cv, n, wl.tau, device=gpu)
    T, K, rounds = 16 * gang, 7, 3
    pk, vk = ap_plonk.Setup(wl.ccs, srs, device=gpu, slots=T)
    ws = batch.WitnessSet(pk, wl.ccs, workloads.variants(wl, K + 1, seed)).to_device().to_pinned(gpu)
    ws.corrupt(K)
    want = oracle_blobs(cv, wl.ccs, srs, ws.items[:K], threads=oracle_threads())
    assert len(set(want)) == K
    for where in ("device", "pinned"):
        pk.paths(reset=True)
        pick = lambda i, r: K if i == 3 else (i + 2 * r) % K
        got, errors = _run_callers(ws, T, rounds, where, pick, expect_error={K: _lib.APK_ERR_WITNESS})
        assert not errors, errors[0]
        assert sum(got.values()) == (T - 1) * rounds
        _assert_every_blob_is_its_own_oracle_proof(got, want, "gangs of %d, %s inputs" % (gang, where))
        p = pk.paths(reset=True)
        assert p["proofs"] == (T - 1) * rounds
        assert p["gang_proofs"] >= p["proofs"] // 2, p                      # most proofs were made as gang members ...
        assert p["gang_msm_launches"] >= 1 and p["gang_ntt_launches"] >= 1 and p["gang_kernel_launches"] >= 1, p
        assert p["msm_batches"] < 4 * p["proofs"], p                        # ... in fewer launch sequences than four per proof
    # alone on the same context: no gang, the latency forms, the same bytes
    pr = _lib.Proof()
    for a in range(K):
        check(ws.prove(a, pr, "device"))
        assert _marshal(pr) == want[a]
    alone = pk.paths(reset=True)
    assert alone["gang_proofs"] == 0 and alone["proofs"] == K, alone
    ws.close()
    pk.close()


def test_ganged_bsb22_proofs(gpu, monkeypatch):
    """Gangs on a circuit with a BSB22 commitment: the Lagrange-basis commitment of every member's own committed column is a merge
    point too (one launch over the Lagrange table for the gang), ahead of the canonical-basis batches."""
    cv, ov = CURVES["bn254"]
    monkeypatch.setenv("APK_GANG", "2")
    seed, log_n = 0xA193, 14
    ccs, w, bl, tau = workloads.random_circuit_bsb22(cv, log_n, seed, nb_commitments=1)
    srs = ap_setup.unsafe_srs(cv, ccs.domain_size(), tau, device=gpu, lagrange=True)
    T, K, rounds = 32, 4, 3
    pk, vk = ap_plonk.Setup(ccs, srs, device=gpu, slots=T)
    vs = [workloads.Variant(w, bl, None, [(0xA193, 0x3910A)])] + workloads.variant_inputs(ccs, K - 1, seed)
    ws = batch.WitnessSet(pk, ccs, vs).to_device()
    want = oracle_blobs(cv, ccs, srs, ws.items, threads=oracle_threads())
    pk.paths(reset=True)
    got, errors = _run_callers(ws, T, rounds, "device", lambda i, r: (i + r) % K)
    assert not errors, errors[0]
    _assert_every_blob_is_its_own_oracle_proof(got, want, "BSB22 in gangs of 2")
    p = pk.paths(reset=True)
    assert p["gang_proofs"] >= p["proofs"] // 2 and p["gang_msm_launches"] >= 1, p
    ws.close()
    pk.close()
