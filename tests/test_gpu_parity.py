"""-m gpu: the HIP path against the oracle on the same seeded inputs, through the C-ABI (bit-exact)."""
import ctypes as C

import pytest

from algoplonk_amd import _lib, ecc, frontend, plonk as ap_plonk, setup as ap_setup
from algoplonk_amd import MarshalProof, MarshalPublicInputs
from algoplonk_amd._lib import lib, check
from oracle import plonk as oplonk
from oracle.prng import SplitMix64, tau_from_seed

from helpers import CURVES, blinding, oracle_circuit_from_ccs, oracle_vk_from_product, random_chain_ccs

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cname", ["bn254", "bls12-381"])
def test_g1_mul_batch(gpu, cname):
    cv, ov = CURVES[cname]
    g = SplitMix64(7)
    scalars = [0, 1, 2, cv.r - 1] + [g.fr(cv.r) for _ in range(60)]
    out = C.create_string_buffer(len(scalars) * 2 * cv.fp_bytes)
    check(lib.apk_g1_mul_batch(cv.abi, gpu, cv.g1_to_bytes(cv.g1), cv.fr_vector(scalars), len(scalars), out))
    got = cv.g1_vector_decode(out.raw)
    for s, P in zip(scalars, got):
        assert P == ov.mul(ov.g1, s)


def _setup_pair(cv, ov, ccs, seed, gpu, lagrange=False, msm_window=0):
    n = ccs.domain_size()
    tau = tau_from_seed(seed, cv.r)
    srs = ap_setup.unsafe_srs(cv, n, tau, device=gpu, lagrange=lagrange)
    pk, vk = ap_plonk.Setup(ccs, srs, device=gpu, msm_window=msm_window)
    osrs = oplonk.synthetic_srs(ov, n, tau, materialize=False)
    opk = oplonk.setup(oracle_circuit_from_ccs(ov, ccs), osrs)
    return pk, vk, opk, srs


@pytest.mark.parametrize("cname", ["bn254", "bls12-381"])
@pytest.mark.parametrize("log_n", [3, 6, 10])
def test_srs_vk_msm_ntt(gpu, cname, log_n):
    cv, ov = CURVES[cname]
    ccs, w, sol = random_chain_ccs(cv, log_n, 0xA190 + log_n)
    pk, vk, opk, srs = _setup_pair(cv, ov, ccs, 11, gpu)
    n = ccs.domain_size()
    tau = srs.tau
    # SRS points = tau^i * G1
    pts = cv.g1_vector_decode(srs.g1)
    assert pts[0] == ov.g1 and pts[1] == ov.mul(ov.g1, tau) and pts[n + 2] == ov.mul(ov.g1, pow(tau, n + 2, cv.r))
    # verifying key = the 8 commitments of plonk.Setup
    ovk = opk.vk
    assert (vk.Ql, vk.Qr, vk.Qm, vk.Qo, vk.Qk) == (ovk.ql, ovk.qr, ovk.qm, ovk.qo, ovk.qk)
    assert vk.S == ovk.s
    assert (vk.SizeInv, vk.Generator, vk.CosetShift) == (ovk.size_inv, ovk.generator, ovk.coset_shift)
    # MSM primitive: random, sparse, all-equal and edge scalars
    g = SplitMix64(5)
    for scalars in ([g.fr(cv.r) for _ in range(n + 3)], [0] * n, [1] * (n + 3), [cv.r - 1] * 5,
                    [g.fr(cv.r) if i % 7 == 0 else 0 for i in range(n)], [3]):
        assert pk.msm(scalars) == ov.mul(ov.g1, oplonk.poly_eval(scalars, tau, cv.r)), len(scalars)
    # NTT primitive, both domains, forward / inverse / coset
    vals = [g.fr(cv.r) for _ in range(n)]
    wn = ov.omega(n)
    assert pk.ntt(vals) == oplonk.ntt(vals, wn, cv.r)
    assert pk.ntt(vals, inverse=True) == oplonk.intt(vals, wn, cv.r)
    v4 = [g.fr(cv.r) for _ in range(4 * n)]
    w4 = ov.omega(4 * n)
    assert pk.ntt(v4, which=1) == oplonk.ntt(v4, w4, cv.r)
    assert pk.ntt(v4, which=1, inverse=True) == oplonk.intt(v4, w4, cv.r)
    u = ov.coset_shift
    shifted = [c * pow(u, i, cv.r) % cv.r for i, c in enumerate(v4)]
    assert pk.ntt(v4, which=1, coset=True) == oplonk.ntt(shifted, w4, cv.r)
    back = pk.ntt(pk.ntt(v4, which=1, coset=True), which=1, inverse=True, coset=True)
    assert back == v4
    pk.close()


@pytest.mark.parametrize("cname", ["bn254", "bls12-381"])
@pytest.mark.parametrize("log_n", [3, 5, 8, 11])
def test_prove_matches_oracle(gpu, cname, log_n):
    cv, ov = CURVES[cname]
    ccs, w, sol = random_chain_ccs(cv, log_n, 0xA190 + 100 * log_n)
    pk, vk, opk, srs = _setup_pair(cv, ov, ccs, 21, gpu)
    bl = blinding(cv, 99)
    proof = ap_plonk.Prove(ccs, pk, w, bl)
    blob = MarshalProof(proof)
    pib = MarshalPublicInputs(w)
    # oracle prover on the same inputs
    oc = oracle_circuit_from_ccs(ov, ccs)
    L, R, O = oplonk.solve_lro(oc, sol)
    tr = oplonk.ProverTrace()
    opr = oplonk.prove(opk, L, R, O, w.public, bl, trace_out=tr)
    ch = proof.challenges
    assert (ch["gamma"], ch["beta"]) == (tr.gamma, tr.beta)
    assert proof.LRO == opr.lro
    assert proof.Z == opr.z and ch["alpha"] == tr.alpha
    assert proof.H == opr.h and ch["zeta"] == tr.zeta
    assert proof.ClaimedValues == opr.claimed_values
    assert proof.ZShiftedOpeningClaimedValue == opr.zshift_value and proof.ZShiftedOpeningH == opr.zshift_h
    assert ch["gamma_kzg"] == tr.gamma_kzg and proof.BatchedProofH == opr.batched_h
    assert blob == oplonk.marshal_proof(ov, opr)
    assert pib == oplonk.marshal_public_inputs(w.public)
    assert len(blob) == (768 if cv is ecc.BN254 else 1056)
    # the verifier transcribed from the reference's template accepts it, and rejects the reference's mutations
    ovk = oracle_vk_from_product(ov, vk)
    assert oplonk.verify(ovk, blob, pib)
    bad = bytearray(pib); bad[-1] ^= 1
    assert not oplonk.verify(ovk, blob, bytes(bad))
    pt = 2 * cv.fp_bytes
    bad = bytearray(blob); bad[:pt] = blob[pt:2 * pt]
    assert not oplonk.verify(ovk, bytes(bad), pib)
    pk.close()
