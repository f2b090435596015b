"""The pin: tests/golden/template_verdicts.json holds what the REFERENCE'S OWN verifier did - the logicsig templates
(/root/reference/verifier/templateLogicSigBN254.go, templateLogicSigBLS12_381.go) rendered the way verifier.WritePythonCode
renders them (/root/reference/verifier/verifier.go:37-122) and executed under an algopy/AVM shim in the build container by
tests/golden/make_template_fixtures.py - on proofs of the reference's test circuits with 0, 1 and 2 BSB22 commitments, both
curves, under the reference's mutations (testutils/verifier_integration_test.go:188-228,232-256).

CPU tier: the oracle's transcription (oracle/plonk.py::verify) and libapk's verifier (apk_verify_ex) must reproduce every
verdict AND every intermediate value (challenges, PI(zeta), lin(zeta), [lin], the folding challenge, the folded opening).
GPU tier: the HIP prover, fed the same circuit, SRS, blinding and BSB22 hiding scalars, must emit exactly the bytes the
template accepted.  Nothing of the reference is read at run time.
"""
import ctypes as C
import json
import os

import pytest

from algoplonk_amd import MarshalProof, MarshalPublicInputs, _lib, ecc, frontend, plonk as ap_plonk, setup as ap_setup
from algoplonk_amd._lib import lib
from oracle import circuits as ocircuits, plonk as oplonk
from oracle.prng import SplitMix64, tau_from_seed

from helpers import CURVES

G = os.path.join(os.path.dirname(__file__), "golden")
FIX = json.load(open(os.path.join(G, "template_verdicts.json")))
CASES = FIX["cases"]
IDS = ["%s-%s" % (c["curve"], c["circuit"]) for c in CASES]


def _pt(j):
    return None if j is None else (int(j[0], 16), int(j[1], 16))


def _oracle_vk(ov, j, pairing=True) -> oplonk.VerifyingKey:
    g2 = tuple(((int(Q[0][0], 16), int(Q[0][1], 16)), (int(Q[1][0], 16), int(Q[1][1], 16))) for Q in j["g2"])
    return oplonk.VerifyingKey(
        curve=ov, size=j["size"], size_inv=int(j["size_inv"], 16), generator=int(j["generator"], 16), coset_shift=j["coset_shift"],
        nb_public=j["nb_public"], ql=_pt(j["ql"]), qr=_pt(j["qr"]), qm=_pt(j["qm"]), qo=_pt(j["qo"]), qk=_pt(j["qk"]),
        s=[_pt(s) for s in j["s"]], qcp=[_pt(q) for q in j["qcp"]],
        commitment_constraint_indexes=list(j["commitment_constraint_indexes"]), g1=_pt(j["g1"]), tau=None, g2=g2 if pairing else None)


def _hexint(v):
    return hex(v) if isinstance(v, int) else v.hex()


def test_the_fixture_names_its_generator_and_the_template_files():
    assert FIX["generator"] == "tests/golden/make_template_fixtures.py"
    assert os.path.exists(os.path.join(G, "make_template_fixtures.py"))
    # all four templates of the reference were rendered and run: the two logicsig verifiers and the two smart-contract ones
    for name in ("templateLogicSigBN254.go", "templateLogicSigBLS12_381.go", "templateSmartContractBN254.go", "templateSmartContractBLS12_381.go"):
        assert len(FIX["templates"][name]["sha256"]) == 64 and len(FIX["templates"][name]["verifier.go_sha256"]) == 64
    # k = 0, 1, 2 commitments on both curves, the valid proof accepted, every mutation rejected
    seen = set()
    for c in CASES:
        seen.add((c["curve"], len(c["vk"]["qcp"])))
        verdicts = {r["mutation"]: r["verdict"] for r in c["results"]}
        assert verdicts.pop("valid") == "accept"
        assert verdicts and set(verdicts.values()) == {"reject"}
        assert {"public_input_byte_flipped", "first_g1_overwritten_by_second", "rekey"} <= set(verdicts)
        # the smart-contract flavour (ARC4 method verify(proof, public_inputs) -> bool) gave the same verdict on every run
        for r in c["results"]:
            assert r["rekey"] or r["smart_contract"]["verdict"] == r["verdict"], (c["circuit"], r["mutation"])
    assert seen == {(cv, k) for cv in ("bn254", "bls12-381") for k in (0, 1, 2)}
    # circuits whose verifying key holds the point at infinity are in: [Qk] (pythagorean), [Qm] (identity, compile_test.go:13-20)
    assert any(c["vk"]["qm"] is None for c in CASES) and any(c["vk"]["qk"] is None for c in CASES)


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_oracle_verifier_reproduces_the_executed_template(case):
    """oracle/plonk.py::verify against the template run: same verdict on every proof, same intermediates wherever both got
    that far (the oracle returns early exactly where the template returns False or the AVM fails the program)."""
    cv, ov = CURVES[case["curve"]]
    vk = _oracle_vk(ov, case["vk"])
    for res in case["results"]:
        if res["rekey"]:
            continue                       # a transaction field, not part of the proof system: nothing to mirror
        T = {}
        ok = oplonk.verify(vk, bytes.fromhex(res["proof"]), bytes.fromhex(res["public_inputs"]), T)
        assert ok == (res["verdict"] == "accept"), (res["mutation"], res["why"])
        want = res["intermediates"]
        common = [k for k in T if k in want]
        if res["verdict"] == "accept" or res["why"] == "returned False" and want:
            # the template ran to its end (or to the pairing): every value the oracle traced must be there and equal
            assert set(T) <= set(want), (res["mutation"], set(T) - set(want))
        for k in common:
            assert _hexint(T[k]) == want[k], (res["mutation"], k)
    # and with the known-tau shortcut instead of the pairing (what the large GPU tests use): same verdicts
    tau = tau_from_seed(case["tau_seed"], cv.r)
    vk_tau = _oracle_vk(ov, case["vk"], pairing=False)
    vk_tau.tau = tau
    for res in case["results"]:
        if not res["rekey"]:
            assert oplonk.verify(vk_tau, bytes.fromhex(res["proof"]), bytes.fromhex(res["public_inputs"])) == (res["verdict"] == "accept")


# ---- libapk's verifier ------------------------------------------------------------------------------------------------

def _g2_bytes(cv, Q):
    return b"".join((int(c, 16) * cv.fp_R % cv.p).to_bytes(cv.fp_bytes, "little") for c in (Q[0][0], Q[0][1], Q[1][0], Q[1][1]))


def _product_vk(cv, j) -> ap_plonk.VerifyingKey:
    return ap_plonk.VerifyingKey(
        curve=cv, Size=j["size"], SizeInv=int(j["size_inv"], 16), Generator=int(j["generator"], 16), CosetShift=j["coset_shift"],
        NbPublicVariables=j["nb_public"], Ql=_pt(j["ql"]), Qr=_pt(j["qr"]), Qm=_pt(j["qm"]), Qo=_pt(j["qo"]), Qk=_pt(j["qk"]),
        S=[_pt(s) for s in j["s"]], Qcp=[_pt(q) for q in j["qcp"]], CommitmentConstraintIndexes=list(j["commitment_constraint_indexes"]),
        KzgG1=_pt(j["g1"]), tau=None, KzgG2=_g2_bytes(cv, j["g2"][0]) + _g2_bytes(cv, j["g2"][1]))


def _raw_proof_from_blob(cv, blob: bytes, k: int) -> _lib.Proof:
    """MarshalProof's layout (helper.go:27-88) back into gnark's in-memory Proof slots."""
    w = 2 * cv.fp_bytes
    p = _lib.Proof()
    p.curve, p.nb_commitments = cv.abi, k

    def pt(slot, off):
        Pt = (int.from_bytes(blob[off: off + w // 2], "big"), int.from_bytes(blob[off + w // 2: off + w], "big"))
        b = cv.g1_to_bytes(Pt)
        C.memmove(slot, b, len(b))

    def fr(slot, off):
        C.memmove(slot, cv.fr_to_mont_bytes(int.from_bytes(blob[off: off + 32], "big")), 32)

    for j in range(3):
        pt(p.lro[j], j * w)
        pt(p.h[j], (3 + j) * w)
    s0 = 6 * w
    for i in range(5):
        fr(p.claimed_values[1 + i], s0 + 32 * i)
    pt(p.z, s0 + 160)
    fr(p.zshift_value, s0 + 160 + w)
    pt(p.batched_h, s0 + 192 + w)
    pt(p.zshift_h, s0 + 192 + 2 * w)
    tail = s0 + 192 + 3 * w
    for i in range(k):
        fr(p.claimed_values[6 + i], tail + 32 * i)
        pt(p.bsb22[i], tail + 32 * k + w * i)
    return p


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_apk_verify_reproduces_the_executed_template(case):
    """libapk's host verifier (the mirror of plonk.Verify at /root/reference/algoplonk.go:93; its own Fp12 tower and ate
    pairing) against the template run: verdicts and the intermediates apk_verify_ex exposes."""
    cv, ov = CURVES[case["curve"]]
    vk = _product_vk(cv, case["vk"])
    k = len(case["vk"]["qcp"])
    names = {"gamma": "gamma", "beta": "beta", "alpha": "alpha", "zeta": "zeta", "pi": "PI", "lin_at_zeta": "linearized_poly_at_z",
             "gamma_kzg": "gamma_kzg", "folded_claim": "folded_claims"}
    for res in case["results"]:
        blob, pib = bytes.fromhex(res["proof"]), bytes.fromhex(res["public_inputs"])
        if res["rekey"] or res["mutation"] in ("claimed_value_plus_r", "proof_truncated_by_one_word"):
            continue       # not expressible in gnark's in-memory Proof (Montgomery limbs below r, fixed slots)
        pub = [int.from_bytes(pib[32 * i: 32 * i + 32], "big") for i in range(len(pib) // 32)]
        raw, rv, tr = _raw_proof_from_blob(cv, blob, k), vk.raw(), _lib.VerifyTrace()
        rc = lib.apk_verify_ex(C.byref(rv), C.byref(raw), cv.fr_vector(pub), len(pub), C.byref(tr))
        assert rc == (0 if res["verdict"] == "accept" else _lib.APK_ERR_VERIFY), (res["mutation"], lib.apk_last_error())
        want = res["intermediates"]
        if res["mutation"] == "z_commitment_off_curve":
            continue       # both refuse the point; the AVM fails at its first use, libapk before the transcript
        for mine, theirs in names.items():
            assert int.from_bytes(bytes(getattr(tr, mine)), "big") == int(want[theirs], 16), (res["mutation"], mine)
        w = 2 * cv.fp_bytes
        assert bytes(tr.lin_commitment)[:w].hex() == want["lin_poly_com"], res["mutation"]
        assert bytes(tr.folded_digest)[:w].hex() == want["folded_digest"], res["mutation"]
    # gnark: len(publicWitness) != vk.NbPublicVariables is an error (never an over-read or a silent truncation)
    res = case["results"][0]
    pib = bytes.fromhex(res["public_inputs"])
    pub = [int.from_bytes(pib[32 * i: 32 * i + 32], "big") for i in range(len(pib) // 32)]
    raw, rv = _raw_proof_from_blob(cv, bytes.fromhex(res["proof"]), k), vk.raw()
    assert lib.apk_verify_ex(C.byref(rv), C.byref(raw), cv.fr_vector(pub + [1]), len(pub) + 1, None) == _lib.APK_ERR_VERIFY
    assert b"invalid witness size" in lib.apk_last_error()
    if pub:
        assert lib.apk_verify_ex(C.byref(rv), C.byref(raw), cv.fr_vector(pub[:-1]), len(pub) - 1, None) == _lib.APK_ERR_VERIFY


def test_plonk_verify_rejects_a_witness_of_the_wrong_length():
    case = CASES[0]
    cv, _ = CURVES[case["curve"]]
    vk = _product_vk(cv, case["vk"])
    res = case["results"][0]
    pib = bytes.fromhex(res["public_inputs"])
    pub = [int.from_bytes(pib[32 * i: 32 * i + 32], "big") for i in range(len(pib) // 32)]
    proof = ap_plonk.Proof(cv, _raw_proof_from_blob(cv, bytes.fromhex(res["proof"]), 0))
    ap_plonk.Verify(proof, vk, frontend.Witness(cv.r, pub, []))
    for bad in (pub[:-1], pub + [5]):
        with pytest.raises(ap_plonk.VerificationError, match="invalid witness size"):
            ap_plonk.Verify(proof, vk, frontend.Witness(cv.r, bad, []))


# ---- GPU tier: the HIP prover emits the accepted bytes -----------------------------------------------------------------

def _ccs_from_oracle_circuit(cv, name):
    """The fixture's circuits (oracle/circuits.py restates the reference's test circuits) as the product's ConstraintSystem,
    with the solver steps gnark's solver would run."""
    ov = CURVES["bn254" if cv is ecc.BN254 else "bls12-381"][1]
    r = cv.r
    if name == "pythagorean":                       # examples/basic/logicsigVerifier/main.go:30-52, assignment (3, 4, 5)
        c, sol = ocircuits.pythagorean(ov)
        solver = [(3, lambda s: s[0] * s[0] % r), (4, lambda s: s[1] * s[1] % r), (5, lambda s: s[2] * s[2] % r), (6, lambda s: (s[3] + s[4]) % r)]
        pub, sec = ["A", "B"], ["C"]
        w = frontend.Witness(r, sol[:2], sol[2:3])
    elif name == "identity":                        # compile_test.go:13-20
        c, sol = ocircuits.identity(ov)
        solver, pub, sec = [], ["X"], []
        w = frontend.Witness(r, sol[:1], [])
    elif name == "random_chain_2p3":
        c, sol = ocircuits.random_chain(ov, 3, 0xA190)
        solver, pub, sec = "gates", ["p0", "p1"], ["s0", "s1"]
        w = frontend.Witness(r, sol[:2], sol[2:4])
    else:                                           # bsb22_test.go:18-39 with 1 / 2 Commit calls
        k = int(name[-1])
        c, sol, plan = ocircuits.bsb22_square(ov, k)
        solver = []
        for i, (cmt, inv) in enumerate(plan):
            solver += [(cmt, ("commit", i)), (inv, ("inv", cmt))]
        pub, sec = ["X"], ["Y"]
        w = frontend.Witness(r, sol[:1], sol[1:2])
    ccs = frontend.ConstraintSystem(r, pub, sec, list(c.constraints), solver, c.nb_variables,
                                    [(list(cm.committed), cm.commitment_index) for cm in c.commitments])
    return ccs, w


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_hip_prover_emits_the_bytes_the_reference_template_accepted(gpu, case):
    cv, ov = CURVES[case["curve"]]
    ccs, w = _ccs_from_oracle_circuit(cv, case["circuit"])
    k = len(ccs.commitments)
    tau = tau_from_seed(case["tau_seed"], cv.r)
    srs = ap_setup.unsafe_srs(cv, ccs.domain_size(), tau, device=gpu, lagrange=k > 0)
    pk, vk = ap_plonk.Setup(ccs, srs, device=gpu)
    j = case["vk"]
    assert [vk.Ql, vk.Qr, vk.Qm, vk.Qo, vk.Qk] == [_pt(j[x]) for x in ("ql", "qr", "qm", "qo", "qk")]
    assert list(vk.S) == [_pt(s) for s in j["s"]] and list(vk.Qcp) == [_pt(q) for q in j["qcp"]]
    assert (vk.Size, vk.SizeInv, vk.Generator, vk.CosetShift) == (j["size"], int(j["size_inv"], 16), int(j["generator"], 16), j["coset_shift"])
    g = SplitMix64(case["blinding_seed"])
    bl = [g.fr(cv.r) for _ in range(9)]
    proof = ap_plonk.Prove(ccs, pk, w, bl, hiding=[tuple(h) for h in case["bsb22_hiding"]] or None)
    valid = case["results"][0]
    assert valid["mutation"] == "valid" and valid["verdict"] == "accept"
    assert MarshalProof(proof).hex() == valid["proof"]
    assert MarshalPublicInputs(w).hex() == valid["public_inputs"]
    want = valid["intermediates"]
    assert {n: hex(v) for n, v in proof.challenges.items()} == {n: want[n] for n in ("gamma", "beta", "alpha", "zeta", "gamma_kzg")}
    # and the library's own verifier, with the G2 side of this SRS, accepts it (Compile -> Verify of the public API does this)
    vk.KzgG2 = ap_setup.g2_from_tau(cv, tau)
    ap_plonk.Verify(proof, vk, w)
    pk.close()
