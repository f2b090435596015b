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
