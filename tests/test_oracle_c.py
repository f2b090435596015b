"""CPU tier: the plain-C oracle (oracle/apk_oracle.c) against the Python oracle and the committed golden vectors."""
import ctypes as C
import json
import os

import pytest

from algoplonk_amd import ecc, frontend
from oracle import c_oracle, plonk as oplonk
from oracle.prng import SplitMix64, tau_from_seed

from helpers import CURVES, blinding, random_chain_ccs

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "proof_vectors.json")))["vectors"]


@pytest.fixture(scope="module")
def clib():
    return c_oracle.load()


@pytest.mark.parametrize("cname", ["bn254", "bls12-381"])
def test_c_msm_and_ntt_match_python(clib, cname):
    cv, ov = CURVES[cname]
    n = 64
    tau = tau_from_seed(5, cv.r)
    srs = oplonk.synthetic_srs(ov, n, tau, materialize=True)
    g = SplitMix64(3)
    for scalars in ([g.fr(cv.r) for _ in range(n + 3)], [0] * n, [1] * n, [cv.r - 1, 0, 5]):
        out = C.create_string_buffer(2 * cv.fp_bytes)
        assert clib.orc_msm(cv.abi, cv.g1_vector(srs.g1), cv.fr_vector(scalars), len(scalars), 2, out) == 0
        assert cv.g1_from_bytes(out.raw) == ov.mul(ov.g1, oplonk.poly_eval(scalars, tau, cv.r))
    vals = [g.fr(cv.r) for _ in range(4 * n)]
    buf = C.create_string_buffer(cv.fr_vector(vals), len(vals) * 32)
    clib.orc_ntt(cv.abi, buf, 4 * n, 0, 0)
    assert cv.fr_vector_decode(buf.raw) == oplonk.ntt(vals, ov.omega(4 * n), cv.r)
    clib.orc_ntt(cv.abi, buf, 4 * n, 1, 0)
    assert cv.fr_vector_decode(buf.raw) == vals


@pytest.mark.parametrize("vec", [v for v in GOLD if v["log_n"] <= 8], ids=lambda v: "%s-2^%d" % (v["curve"], v["log_n"]))
def test_c_prover_reproduces_golden_vectors(clib, vec):
    cv, ov = CURVES[vec["curve"]]
    ccs, w, sol = random_chain_ccs(cv, vec["log_n"], vec["circuit_seed"])
    n = ccs.domain_size()
    srs = oplonk.synthetic_srs(ov, n, tau_from_seed(vec["tau_seed"], cv.r), materialize=True)
    tr = frontend.build_trace(ccs)
    L, R, O = frontend.wire_columns(ccs, sol)
    rc, blob, ch = c_oracle.prove(clib, cv.abi, n, ccs.GetNbPublicVariables(), cv.g1_vector(srs.g1),
                                  [cv.fr_vector(x) for x in (tr.ql, tr.qr, tr.qm, tr.qo, tr.qk)], tr.perm,
                                  cv.fr_vector(L), cv.fr_vector(R), cv.fr_vector(O), cv.fr_vector(w.public),
                                  cv.fr_vector(blinding(cv, vec["blinding_seed"])), threads=2)
    assert rc == 0
    assert blob.hex() == vec["proof"]
    assert [hex(x) for x in ch] == [vec["challenges"][k] for k in ("gamma", "beta", "alpha", "zeta", "gamma_kzg")]


def test_c_prover_flags_unsatisfied_witness(clib):
    cv, ov = CURVES["bn254"]
    ccs, w, sol = random_chain_ccs(cv, 4, 1)
    n = ccs.domain_size()
    srs = oplonk.synthetic_srs(ov, n, 9, materialize=True)
    tr = frontend.build_trace(ccs)
    L, R, O = frontend.wire_columns(ccs, sol)
    O[5] = (O[5] + 1) % cv.r
    rc, _, _ = c_oracle.prove(clib, cv.abi, n, ccs.GetNbPublicVariables(), cv.g1_vector(srs.g1),
                              [cv.fr_vector(x) for x in (tr.ql, tr.qr, tr.qm, tr.qo, tr.qk)], tr.perm, cv.fr_vector(L),
                              cv.fr_vector(R), cv.fr_vector(O), cv.fr_vector(w.public), cv.fr_vector(blinding(cv, 1)))
    assert rc == 4


@pytest.mark.parametrize("cname", ["bn254", "bls12-381"])
def test_fast_msm_matches_the_plain_one(clib, cname):
    """oracle/fast_msm_tmpl.h (batch-affine Pippenger, signed digits, thread pool - bench.py's cpu_baseline) against orc_msm, byte
    for byte: uniform scalars, the skewed inputs that drive its conflict queue (all equal: every point of a window in ONE bucket),
    doublings and cancellations (repeated bases with equal / opposite scalars), zeros, lengths that leave chunks ragged."""
    cv, ov = CURVES[cname]
    n = 600
    tau = tau_from_seed(5, cv.r)
    srs = oplonk.synthetic_srs(ov, 1024, tau, materialize=True)
    pts = list(srs.g1[:n])
    g = SplitMix64(11)
    cases = {"uniform": (pts, [g.fr(cv.r) for _ in range(n)]), "ones": (pts, [1] * n), "minus_one": (pts, [cv.r - 1] * n),
             "zeros_mostly": (pts, [0 if i % 7 else g.fr(cv.r) for i in range(n)]), "short": (pts[:3], [5, 0, cv.r - 2]),
             "same_base_doubling": ([pts[1]] * 64, [3] * 64), "same_base_cancel": ([pts[1]] * 64, [3 if i % 2 else cv.r - 3 for i in range(64)]),
             "small": (pts, [g.below(1 << 12) for _ in range(n)])}
    for name, (bases, sc) in cases.items():
        want, got = C.create_string_buffer(2 * cv.fp_bytes), C.create_string_buffer(2 * cv.fp_bytes)
        b, s = cv.g1_vector(bases), cv.fr_vector(sc)
        assert clib.orc_msm(cv.abi, b, s, len(sc), 2, want) == 0
        for threads in (1, 3, 8):
            assert clib.orc_msm_fast(cv.abi, b, s, len(sc), threads, got) == 0
            assert got.raw == want.raw, (name, threads)


@pytest.mark.parametrize("cname,log_n", [("bn254", 6), ("bn254", 11), ("bls12-381", 5), ("bls12-381", 10)])
def test_fast_prover_gives_the_bytes_of_the_plain_one(clib, cname, log_n):
    """oracle/fast_prover.c - circuit-only work once, parallel FFTs, batch-affine MSMs, quotient on one coset, [lin] from
    commitments, lin(zeta) from the verifier's identity - must return orc_prove's proof and challenges (and through the test
    above this one, the golden vectors' and the Python oracle's), at several thread counts and on repeated use of one context."""
    cv, ov = CURVES[cname]
    ccs, w, sol = random_chain_ccs(cv, log_n, 41 + log_n)
    n = ccs.domain_size()
    srs = oplonk.synthetic_srs(ov, n, tau_from_seed(9, cv.r), materialize=True)
    tr = frontend.build_trace(ccs)
    L, R, O = frontend.wire_columns(ccs, sol)
    cols = [cv.fr_vector(x) for x in (tr.ql, tr.qr, tr.qm, tr.qo, tr.qk)]
    args = (cv.fr_vector(L), cv.fr_vector(R), cv.fr_vector(O), cv.fr_vector(w.public))
    fp = c_oracle.FastProver(clib, cv.abi, n, ccs.GetNbPublicVariables(), cv.g1_vector(srs.g1), cols, tr.perm, threads=4)
    for seed, threads in ((1, 1), (2, 4), (3, 7)):
        bl = cv.fr_vector(blinding(cv, seed))
        rc, want, wch = c_oracle.prove(clib, cv.abi, n, ccs.GetNbPublicVariables(), cv.g1_vector(srs.g1), cols, tr.perm, *args, bl, threads=2)
        rc2, got, gch = fp.prove(*args, bl, threads=threads)
        assert rc == 0 and rc2 == 0 and got == want and gch == wch, (seed, threads)
    # an unsatisfied witness is reported, like orc_prove does
    bad = list(O)
    bad[5] = (bad[5] + 1) % cv.r
    assert fp.prove(cv.fr_vector(L), cv.fr_vector(R), cv.fr_vector(bad), cv.fr_vector(w.public), cv.fr_vector(blinding(cv, 1)), threads=2)[0] == 4
    fp.close()


@pytest.mark.parametrize("cname,nb", [("bn254", 1), ("bn254", 2), ("bls12-381", 1), ("bls12-381", 2)])
def test_fast_prover_bsb22_matches_the_python_oracle(clib, cname, nb):
    """The BSB22 path of oracle/fast_prover.c (commitment over the Lagrange SRS, hash_fr into Qk, Qcp * pi2 in the quotient, the
    extra terms of lin / the fold / the transcripts, the blob tail of helper.go:74-85) against oracle/plonk.py on the reference's
    own commitment circuit (bsb22_test.go:18-39) - the C oracle proper has no BSB22 path, so this is what lets bench.py compare
    the GPU's BASELINE configs[4] proofs with a host prover at full size."""
    from oracle import circuits as ocircuits
    cv, ov = CURVES[cname]
    c, sol0, plan = ocircuits.bsb22_square(ov, nb)
    n = c.domain_size()
    tau = tau_from_seed(23, cv.r)
    osrs = oplonk.synthetic_srs(ov, n, tau, materialize=True)
    opk = oplonk.setup(c, osrs)
    g = SplitMix64(5)
    hiding = [(g.fr(cv.r), g.fr(cv.r)) for _ in range(nb)]
    wn = ov.omega(n)
    sol, pi2 = ocircuits.solve_bsb22(c, sol0, plan, lambda col: osrs.commit(oplonk.intt(col, wn, cv.r)), hiding)
    L, R, O = oplonk.solve_lro(c, sol)
    public = sol[: c.nb_public]
    bl = blinding(cv, 3)
    want = oplonk.marshal_proof(ov, oplonk.prove(opk, L, R, O, public, bl, pi2=pi2))
    # Lagrange SRS: [L_i(tau)]G1
    lag_scalars = [oplonk.poly_eval(oplonk.intt([1 if j == i else 0 for j in range(n)], wn, cv.r), tau, cv.r) for i in range(n)]
    lag = cv.g1_vector([ov.mul(ov.g1, s) for s in lag_scalars])
    tr = opk.trace
    cols = [cv.fr_vector(x) for x in (tr.ql, tr.qr, tr.qm, tr.qo, tr.qk)]
    fp = c_oracle.FastProver(clib, cv.abi, n, c.nb_public, cv.g1_vector(osrs.g1), cols, list(tr.S), threads=3,
                             qcp=[cv.fr_vector(q) for q in tr.qcp], cci=list(opk.vk.commitment_constraint_indexes), srs_lagrange=lag)
    rc, got, _ = fp.prove(cv.fr_vector(L), cv.fr_vector(R), cv.fr_vector(O), cv.fr_vector(public), cv.fr_vector(bl), threads=3,
                          pi2=[cv.fr_vector(p) for p in pi2])
    assert rc == 0 and got == want
    fp.close()
