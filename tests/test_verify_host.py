"""CPU tier: libapk's host-side verifier (apk_verify = the mirror of gnark's plonk.Verify, /root/reference/algoplonk.go:93)
against the oracle.  The oracle's verifier is a transcription of the reference's AVM template with its own plain-Python
pairing; apk_verify is written from the identity on a C++ tower with the ate pairing: two independent
implementations that must accept and reject the same proofs."""
import ctypes as C
import os

import pytest

from algoplonk_amd import _lib, ecc, plonk as ap_plonk, setup as ap_setup
from algoplonk_amd._lib import lib, check
from oracle import circuits as ocircuits, curves as oc, plonk as oplonk
from oracle.prng import SplitMix64, tau_from_seed

from helpers import CURVES

G = os.path.join(os.path.dirname(__file__), "golden")


def _pairing(cname):
    if cname == "bn254":
        from oracle import pairing_bn254 as pr
    else:
        from oracle import pairing_bls12381 as pr
    return pr


def _g2_bytes(cv, Q):
    """oracle G2 point ((x0, x1), (y0, y1)) -> gnark in-memory G2Affine bytes"""
    R = cv.fp_R
    return b"".join((c * R % cv.p).to_bytes(cv.fp_bytes, "little") for c in (Q[0][0], Q[0][1], Q[1][0], Q[1][1]))


def _product_vk(cv, ovk, g2_bytes) -> ap_plonk.VerifyingKey:
    return ap_plonk.VerifyingKey(curve=cv, Size=ovk.size, SizeInv=ovk.size_inv, Generator=ovk.generator, CosetShift=ovk.coset_shift,
                                 NbPublicVariables=ovk.nb_public, Ql=ovk.ql, Qr=ovk.qr, Qm=ovk.qm, Qo=ovk.qo, Qk=ovk.qk, S=list(ovk.s),
                                 Qcp=list(ovk.qcp), CommitmentConstraintIndexes=list(ovk.commitment_constraint_indexes), KzgG1=ovk.g1,
                                 tau=None, KzgG2=g2_bytes)


def _raw_proof(cv, opr) -> _lib.Proof:
    """oracle Proof -> apk_proof (gnark in-memory slots)"""
    p = _lib.Proof()
    p.curve, p.nb_commitments = cv.abi, len(opr.bsb22_commitments)

    def pt(slot, P):
        b = cv.g1_to_bytes(P)
        C.memmove(slot, b, len(b))

    def fr(slot, x):
        C.memmove(slot, cv.fr_to_mont_bytes(x), 32)

    for j in range(3):
        pt(p.lro[j], opr.lro[j]); pt(p.h[j], opr.h[j])
    pt(p.z, opr.z); pt(p.batched_h, opr.batched_h); pt(p.zshift_h, opr.zshift_h)
    for k, P in enumerate(opr.bsb22_commitments):
        pt(p.bsb22[k], P)
    for i, v in enumerate(opr.claimed_values):
        fr(p.claimed_values[i], v)
    fr(p.zshift_value, opr.zshift_value)
    return p


def _verify(vk: ap_plonk.VerifyingKey, raw, public) -> int:
    rv = vk.raw()
    return lib.apk_verify(C.byref(rv), C.byref(raw), vk.curve.fr_vector(public))


@pytest.mark.parametrize("cname", ["bn254", "bls12-381"])
def test_g2_encodings_against_the_oracle_and_the_reference_files(cname):
    """vk.bin of the reference's trusted setups (tests/golden/*.vk.bin = G2[0] || G2[1] || G1[0], compressed): apk_g2_decompress
    must give the points the oracle's decoder gives, G2[0] must be the generator the library multiplies
    (setup/trusted_setup_test.go:33-39,93-96,220-223), and [k]G2 must match the oracle's scalar multiplication."""
    cv, ov = CURVES[cname]
    pr = _pairing(cname)
    names = ["PerpetualPowersOfTauBN254"] if cname == "bn254" else ["EethereumKzgCeremonyBLS12_381", "DuskBLS12_381"]
    gen = C.create_string_buffer(4 * cv.fp_bytes)
    check(lib.apk_g2_mul_generator(cv.abi, cv.fr_vector([1]), gen))
    assert gen.raw == _g2_bytes(cv, pr.G2_GEN)
    for name in names:
        vkb = open(os.path.join(G, name + ".vk.bin"), "rb").read()
        g2 = ap_setup.g2_from_vk_bin(cv, vkb)
        w = 2 * cv.fp_bytes
        want = [pr.g2_decompress(vkb[:w]), pr.g2_decompress(vkb[w: 2 * w])]
        assert g2 == _g2_bytes(cv, want[0]) + _g2_bytes(cv, want[1])
        assert g2[: 4 * cv.fp_bytes] == gen.raw
    k = SplitMix64(3).fr(cv.r)
    out = C.create_string_buffer(4 * cv.fp_bytes)
    check(lib.apk_g2_mul_generator(cv.abi, cv.fr_vector([k]), out))
    assert out.raw == _g2_bytes(cv, pr.g2_mul(pr.G2_GEN, k))
    bad = bytearray(open(os.path.join(G, names[0] + ".vk.bin"), "rb").read()[: 2 * cv.fp_bytes])
    bad[0] &= 0x1F                                                    # flag bits cleared: not a compressed encoding
    assert lib.apk_g2_decompress(cv.abi, bytes(bad), out) == _lib.APK_ERR_ARG


@pytest.mark.parametrize("cname", ["bn254", "bls12-381"])
def test_apk_verify_accepts_and_rejects_what_the_transcribed_verifier_does(cname):
    """Oracle proofs of the reference's test circuits (Pythagorean 3,4,5: examples/basic/logicsigVerifier/main.go:30-52;
    identity: compile_test.go:13-20; X == Y*Y: bsb22_test.go:18-39; a random chain) through apk_verify, with the reference's
    mutations (testutils/verifier_integration_test.go:188-228): flipped public input, first G1 := second G1, flipped scalar."""
    cv, ov = CURVES[cname]
    tau = tau_from_seed(0x7E57, cv.r)
    g2 = ap_setup.g2_from_tau(cv, tau)
    for name, (c, sol) in {"pyth": ocircuits.pythagorean(ov), "id": ocircuits.identity(ov), "sq": ocircuits.square(ov),
                           "rnd": ocircuits.random_chain(ov, 4, 0xA190)}.items():
        opk = oplonk.setup(c, oplonk.synthetic_srs(ov, c.domain_size(), tau, materialize=False))
        L, R, O = oplonk.solve_lro(c, sol)
        pub = sol[: c.nb_public]
        g = SplitMix64(5)
        opr = oplonk.prove(opk, L, R, O, pub, [g.fr(cv.r) for _ in range(9)])
        assert oplonk.verify(opk.vk, oplonk.marshal_proof(ov, opr), oplonk.marshal_public_inputs(pub))
        vk = _product_vk(cv, opk.vk, g2)
        raw = _raw_proof(cv, opr)
        assert _verify(vk, raw, pub) == 0, (name, lib.apk_last_error())
        # the lin(zeta) slot is recomputed by the verifier, never trusted (it is not even in the AVM blob: helper.go:53-56)
        junk = _raw_proof(cv, opr); C.memmove(junk.claimed_values[0], cv.fr_to_mont_bytes(12345), 32)
        assert _verify(vk, junk, pub) == 0
        if pub:
            assert _verify(vk, raw, [(pub[0] + 1) % cv.r] + pub[1:]) == _lib.APK_ERR_VERIFY
        bad = _raw_proof(cv, opr); C.memmove(bad.lro[0], bytes(bad.lro[1]), 96)
        assert _verify(vk, bad, pub) == _lib.APK_ERR_VERIFY
        bad = _raw_proof(cv, opr); C.memmove(bad.claimed_values[2], cv.fr_to_mont_bytes((opr.claimed_values[2] + 1) % cv.r), 32)
        assert _verify(vk, bad, pub) == _lib.APK_ERR_VERIFY
        bad = _raw_proof(cv, opr); C.memmove(bad.zshift_h, bytes(bad.batched_h), 96)
        assert _verify(vk, bad, pub) == _lib.APK_ERR_VERIFY
        bad = _raw_proof(cv, opr); bad.z[0] ^= 1                       # no longer a curve point
        assert _verify(vk, bad, pub) == _lib.APK_ERR_VERIFY
        # a key for another tau rejects everything
        assert _verify(_product_vk(cv, opk.vk, ap_setup.g2_from_tau(cv, tau + 1)), raw, pub) == _lib.APK_ERR_VERIFY


@pytest.mark.parametrize("cname", ["bn254", "bls12-381"])
@pytest.mark.parametrize("k", [1, 2])
def test_apk_verify_with_bsb22_commitments(cname, k):
    """bsb22_test.go:46-123 circuits: the hash_fr / L_{nbPublic+cci}(zeta) term of PI(zeta) and the qcp / [pi2] terms."""
    cv, ov = CURVES[cname]
    tau = tau_from_seed(0xB5B, cv.r)
    c, sol, plan = ocircuits.bsb22_square(ov, k)
    n = c.domain_size()
    osrs = oplonk.synthetic_srs(ov, n, tau, materialize=False)
    opk = oplonk.setup(c, osrs)
    wn = ov.omega(n)
    sol, pi2 = ocircuits.solve_bsb22(c, sol, plan, lambda col: osrs.commit(oplonk.intt(col, wn, cv.r)), [(11 + i, 22 + i) for i in range(k)])
    L, R, O = oplonk.solve_lro(c, sol)
    opr = oplonk.prove(opk, L, R, O, sol[:1], list(range(31, 40)), pi2=pi2)
    assert oplonk.verify(opk.vk, oplonk.marshal_proof(ov, opr), oplonk.marshal_public_inputs(sol[:1]))
    vk = _product_vk(cv, opk.vk, ap_setup.g2_from_tau(cv, tau))
    raw = _raw_proof(cv, opr)
    assert _verify(vk, raw, sol[:1]) == 0, lib.apk_last_error()
    bad = _raw_proof(cv, opr); C.memmove(bad.bsb22[k - 1], bytes(bad.lro[0]), 96)      # another commitment point
    assert _verify(vk, bad, sol[:1]) == _lib.APK_ERR_VERIFY
    bad = _raw_proof(cv, opr); C.memmove(bad.claimed_values[6], cv.fr_to_mont_bytes(7), 32)   # qcp_0(zeta)
    assert _verify(vk, bad, sol[:1]) == _lib.APK_ERR_VERIFY
    wrong = _product_vk(cv, opk.vk, ap_setup.g2_from_tau(cv, tau))
    wrong.CommitmentConstraintIndexes[0] += 1
    assert _verify(wrong, raw, sol[:1]) == _lib.APK_ERR_VERIFY


def test_apk_verify_under_the_real_ethereum_ceremony_key():
    """A proof under the REAL Ethereum KZG ceremony SRS (nobody knows tau): the C oracle proves a 2^5 circuit over the first
    35 points of the reference's pk.bin, and apk_verify accepts it against the ceremony's own G2 points (vk.bin)."""
    from oracle import c_oracle
    cv, ov = CURVES["bls12-381"]
    c, sol = ocircuits.random_chain(ov, 5, 0xA191)
    n = c.domain_size()
    pkb = open(os.path.join(G, "setup", "EethereumKzgCeremonyBLS12_381", "pk.bin"), "rb").read()
    pts = [ov.decompress(pkb[4 + 48 * i: 4 + 48 * (i + 1)]) for i in range(n + 3)]
    opk = oplonk.setup(c, oplonk.SRS(ov, pts, [], None))             # VK commitments by naive MSM over the ceremony points
    tr = opk.trace
    L, R, O = oplonk.solve_lro(c, sol)
    pub = sol[: c.nb_public]
    g = SplitMix64(8)
    bl = [g.fr(cv.r) for _ in range(9)]
    rc, blob, _ = c_oracle.prove(c_oracle.load(), cv.abi, n, c.nb_public, cv.g1_vector(pts),
                                 [cv.fr_vector(x) for x in (tr.ql, tr.qr, tr.qm, tr.qo, tr.qk)], tr.S, cv.fr_vector(L), cv.fr_vector(R),
                                 cv.fr_vector(O), cv.fr_vector(pub), cv.fr_vector(bl), threads=4)
    assert rc == 0 and len(blob) == 1056
    # blob (helper.go:27-88 layout) -> apk_proof
    P = lambda off: (int.from_bytes(blob[off: off + 48], "big"), int.from_bytes(blob[off + 48: off + 96], "big"))
    S = lambda off: int.from_bytes(blob[off: off + 32], "big")
    opr = oplonk.Proof(lro=[P(0), P(96), P(192)], h=[P(288), P(384), P(480)], claimed_values=[0] + [S(576 + 32 * i) for i in range(5)],
                       z=P(736), zshift_value=S(832), batched_h=P(864), zshift_h=P(960), bsb22_commitments=[])
    assert oplonk.marshal_proof(ov, opr) == blob
    vkb = open(os.path.join(G, "EethereumKzgCeremonyBLS12_381.vk.bin"), "rb").read()
    vk = _product_vk(cv, opk.vk, ap_setup.g2_from_vk_bin(cv, vkb))
    raw = _raw_proof(cv, opr)
    assert _verify(vk, raw, pub) == 0, lib.apk_last_error()
    assert _verify(vk, raw, [pub[0] ^ 1] + pub[1:]) == _lib.APK_ERR_VERIFY
