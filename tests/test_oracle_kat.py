"""CPU tier: pin the oracle against every known answer the reference's own tests hold for this path:
SRS file parsing + point decompression (setup/trusted_setup_test.go), proof blob shape (bsb22_test.go:70,83,97-120),
verifier accept / reject behaviour (testutils/verifier_integration_test.go:188-228)."""
import json
import os

import pytest

from oracle import circuits as ocircuits, curves as oc, plonk as oplonk
from oracle.prng import SplitMix64, tau_from_seed

G = os.path.join(os.path.dirname(__file__), "golden")
KAT = json.load(open(os.path.join(G, "trusted_setup_kat.json")))


def test_ethereum_srs_head_decompresses_to_reference_points():
    """setup/trusted_setup_test.go:172-288 (first five G1, G1[0] = generator, G1[32767])."""
    cv = oc.BLS12_381
    head = open(os.path.join(G, "EethereumKzgCeremonyBLS12_381.pk.head.bin"), "rb").read()
    assert int.from_bytes(head[:4], "big") == 32768          # loadTrustedSetupBytes reads this count (setup.go:216)
    for i, want_hex in enumerate(KAT["ethereum_g1_first5"]):
        raw = head[4 + 48 * i: 4 + 48 * (i + 1)]
        assert raw.hex() == want_hex
        P = cv.decompress(raw)
        assert cv.is_on_curve(P)
        # the reference re-derives X by clearing the three flag bits (zeroFirstThreeBits)
        assert P[0] == int(want_hex, 16) & ((1 << 381) - 1)
        assert cv.compress(P) == raw
    assert cv.decompress(head[4:52]) == cv.g1
    last = open(os.path.join(G, "EethereumKzgCeremonyBLS12_381.pk.32767.bin"), "rb").read()
    assert last.hex() == KAT["ethereum_g1_32767"]
    assert cv.is_on_curve(cv.decompress(last))


def test_dusk_known_answers_decompress():
    """setup/trusted_setup_test.go:53-59,132: Dusk's pk.bin is not mounted, but its KAT hex strings are data."""
    cv = oc.BLS12_381
    pts = [cv.decompress(bytes.fromhex(h)) for h in KAT["dusk_g1_first5"] + [KAT["dusk_g1_32767"]]]
    assert all(cv.is_on_curve(P) for P in pts) and pts[0] == cv.g1
    # both ceremonies start [1]G1, [tau]G1 with different tau
    assert KAT["dusk_g1_first5"][0] == KAT["ethereum_g1_first5"][0] and KAT["dusk_g1_first5"][1] != KAT["ethereum_g1_first5"][1]


@pytest.mark.parametrize("name,cv", [("PerpetualPowersOfTauBN254", oc.BN254), ("EethereumKzgCeremonyBLS12_381", oc.BLS12_381),
                                      ("DuskBLS12_381", oc.BLS12_381)])
def test_vk_bin_layout(name, cv):
    """vk.bin = G2[0] || G2[1] || G1[0] compressed; Vk.G1 == G1 generator
    (setup/trusted_setup_test.go:33-36,127-129)."""
    b = open(os.path.join(G, name + ".vk.bin"), "rb").read()
    n = cv.fp_bytes
    assert len(b) == 5 * n
    assert cv.decompress(b[4 * n:]) == cv.g1


@pytest.mark.parametrize("cv", [oc.BN254, oc.BLS12_381])
def test_prover_verifier_roundtrip_and_mutations(cv):
    """examples/basic Pythagorean circuit (3,4,5), compile_test identity, X == Y*Y, a random chain."""
    for name, (c, sol) in {"pyth": ocircuits.pythagorean(cv), "id": ocircuits.identity(cv), "sq": ocircuits.square(cv),
                           "rnd": ocircuits.random_chain(cv, 5, 0xA190)}.items():
        n = c.domain_size()
        pk = oplonk.setup(c, oplonk.synthetic_srs(cv, n, tau_from_seed(1, cv.r), materialize=False))
        L, R, O = oplonk.solve_lro(c, sol)
        pub = sol[: c.nb_public]
        assert oplonk.check_gates(c, pk.trace, L, R, O, pub)
        g = SplitMix64(99)
        pr = oplonk.prove(pk, L, R, O, pub, [g.fr(cv.r) for _ in range(9)])
        blob, pib = oplonk.marshal_proof(cv, pr), oplonk.marshal_public_inputs(pub)
        assert len(blob) == (24 * 32 if cv is oc.BN254 else 33 * 32)        # bsb22_test.go:70,83 base words
        assert len(pib) == 32 * c.nb_public                                 # templateLogicSigBN254.go:51
        assert oplonk.verify(pk.vk, blob, pib), name
        bad = bytearray(pib); bad[-1] ^= 1                                  # verifier_integration_test.go:199-213
        assert not oplonk.verify(pk.vk, blob, bytes(bad))
        pt = 2 * cv.fp_bytes
        bad = bytearray(blob); bad[:pt] = blob[pt: 2 * pt]                  # :215-228 first G1 := second G1
        assert not oplonk.verify(pk.vk, bytes(bad), pib)
        assert not oplonk.verify(pk.vk, blob[:-1], pib)                     # length check :50


@pytest.mark.parametrize("cv", [oc.BN254, oc.BLS12_381])
def test_unsatisfied_witness_is_rejected_by_the_oracle_prover(cv):
    c, sol = ocircuits.pythagorean(cv, 3, 4, 6)
    pk = oplonk.setup(c, oplonk.synthetic_srs(cv, c.domain_size(), 5, materialize=False))
    L, R, O = oplonk.solve_lro(c, sol)
    with pytest.raises(AssertionError):
        oplonk.prove(pk, L, R, O, sol[:2], list(range(1, 10)))


@pytest.mark.parametrize("cv", [oc.BN254, oc.BLS12_381])
def test_msm_definition_and_known_tau_shortcut_agree(cv):
    n = 8
    tau = tau_from_seed(3, cv.r)
    srs = oplonk.synthetic_srs(cv, n, tau, materialize=True)
    g = SplitMix64(1)
    coeffs = [g.fr(cv.r) for _ in range(n + 3)]
    assert cv.msm_naive(srs.g1, coeffs) == cv.mul(cv.g1, oplonk.poly_eval(coeffs, tau, cv.r))
    evals = oplonk.ntt(coeffs[:n], cv.omega(n), cv.r)
    assert srs.commit_lagrange(evals) == cv.mul(cv.g1, oplonk.poly_eval(coeffs[:n], tau, cv.r))   # Lagrange SRS = canonical commit


def test_hash_fr_matches_expand_msg_xmd_shape():
    """templateLogicSigBN254.go:386-397: 48 bytes of expand_msg_xmd reduced mod r."""
    v = oplonk.hash_fr(oc.BN254.raw_bytes(oc.BN254.g1), oc.BN254.r)
    assert 0 <= v < oc.BN254.r and v != oplonk.hash_fr(oc.BN254.raw_bytes(oc.BN254.mul(oc.BN254.g1, 2)), oc.BN254.r)


def test_pairing_and_g2_decoding_against_reference_expectations():
    """setup/trusted_setup_test.go:194-253: vk.bin's first G2 point is the BLS12-381 G2 generator; plus the defining
    relation of any KZG SRS, e([tau]G1, G2) == e(G1, [tau]G2), on the REAL Ethereum ceremony data."""
    from oracle import pairing_bls12381 as pr
    cv = oc.BLS12_381
    vk = open(os.path.join(G, "EethereumKzgCeremonyBLS12_381.vk.bin"), "rb").read()
    g20, g21 = pr.g2_decompress(vk[:96]), pr.g2_decompress(vk[96:192])
    assert g20 == pr.G2_GEN and pr.g2_on_curve(g21)
    head = open(os.path.join(G, "EethereumKzgCeremonyBLS12_381.pk.head.bin"), "rb").read()
    tau_g1 = cv.decompress(head[4 + 48: 4 + 96])
    assert pr.pairing_check([tau_g1, cv.neg(cv.g1)], [g20, g21])
    assert not pr.pairing_check([tau_g1, cv.neg(cv.mul(cv.g1, 2))], [g20, g21])


def test_bn254_g2_decoding_and_pairing():
    """setup/trusted_setup_test.go:22-40: the PPoT vk.bin starts with the BN254 G2 generator; bilinearity of the pairing."""
    from oracle import pairing_bn254 as pr
    cv = oc.BN254
    vk = open(os.path.join(G, "PerpetualPowersOfTauBN254.vk.bin"), "rb").read()
    g20, g21 = pr.g2_decompress(vk[:64]), pr.g2_decompress(vk[64:128])
    assert g20 == pr.G2_GEN and pr.g2_on_curve(g21) and g21 != g20
    assert pr.pairing_check([cv.mul(cv.g1, 7), cv.neg(cv.mul(cv.g1, 77))], [pr.g2_mul(pr.G2_GEN, 11), pr.G2_GEN])
    assert not pr.pairing_check([cv.mul(cv.g1, 7), cv.neg(cv.mul(cv.g1, 78))], [pr.g2_mul(pr.G2_GEN, 11), pr.G2_GEN])


@pytest.mark.parametrize("cv", [oc.BN254, oc.BLS12_381])
def test_verifier_with_real_pairing_agrees_with_known_tau_shortcut(cv):
    """The transcription's last line, ec.pairing_check (templateLogicSigBN254.go:350-355,
    templateLogicSigBLS12_381.go:366-371), run for real."""
    if cv is oc.BN254:
        from oracle import pairing_bn254 as pr
    else:
        from oracle import pairing_bls12381 as pr
    import dataclasses
    c, sol = ocircuits.pythagorean(cv)
    tau = tau_from_seed(77, cv.r)
    pk = oplonk.setup(c, oplonk.synthetic_srs(cv, c.domain_size(), tau, materialize=False))
    L, R, O = oplonk.solve_lro(c, sol)
    pr_ = oplonk.prove(pk, L, R, O, sol[:2], list(range(11, 20)))
    blob, pib = oplonk.marshal_proof(cv, pr_), oplonk.marshal_public_inputs(sol[:2])
    vk_pairing = dataclasses.replace(pk.vk, tau=None, g2=(pr.G2_GEN, pr.g2_mul(pr.G2_GEN, tau)))
    assert oplonk.verify(pk.vk, blob, pib) and oplonk.verify(vk_pairing, blob, pib)
    bad = bytearray(blob); bad[700] ^= 1
    assert not oplonk.verify(vk_pairing, bytes(bad), pib)
