"""-m gpu: the setup-time GPU work of SURVEY.md §8f.1 (row a12) - SRS decompression, ToLagrangeG1, and BASELINE.json
configs[2]: a BLS12-381 2^14 circuit proved under the REAL Ethereum KZG ceremony SRS (no tau known to anyone)."""
import ctypes as C
import hashlib
import json
import os

import pytest

from algoplonk_amd import _lib, ecc, frontend, plonk as ap_plonk, setup as ap_setup, workloads
from algoplonk_amd import MarshalProof, MarshalPublicInputs
from algoplonk_amd._lib import lib, check
from oracle import c_oracle, plonk as oplonk
from oracle.prng import SplitMix64, tau_from_seed

from helpers import CURVES, oracle_threads

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
KAT = json.load(open(os.path.join(G, "trusted_setup_kat.json")))


@pytest.mark.parametrize("cname", ["bn254", "bls12-381"])
def test_decompress_matches_oracle_and_rejects_bad_points(gpu, cname):
    cv, ov = CURVES[cname]
    g = SplitMix64(3)
    pts = [ov.mul(ov.g1, g.fr(cv.r)) for _ in range(40)] + [None, ov.g1, ov.neg(ov.g1)]
    comp = b"".join(ov.compress(P) for P in pts)
    got = cv.g1_vector_decode(ap_setup.decompress_g1_batch(cv, comp, gpu))
    assert got == pts
    # an x that is not on the curve, and a missing compression flag
    x = 5
    while oplonk_sqrt(ov, x) is not None:
        x += 1
    bad = bytearray(x.to_bytes(cv.fp_bytes, "big")); bad[0] |= 0x80
    with pytest.raises(_lib.ApkError, match="not valid G1"):
        ap_setup.decompress_g1_batch(cv, bytes(bad), gpu)
    with pytest.raises(_lib.ApkError):
        ap_setup.decompress_g1_batch(cv, ov.g1[0].to_bytes(cv.fp_bytes, "big"), gpu)   # flags 00


def oplonk_sqrt(ov, x):
    from oracle.curves import sqrt_mod
    return sqrt_mod((x * x * x + ov.b) % ov.p, ov.p)


def test_decompress_reference_known_answers(gpu):
    """setup/trusted_setup_test.go:172-288 through the GPU path."""
    cv, ov = CURVES["bls12-381"]
    head = open(os.path.join(G, "EethereumKzgCeremonyBLS12_381.pk.head.bin"), "rb").read()[4:]
    pts = cv.g1_vector_decode(ap_setup.decompress_g1_batch(cv, head, gpu))
    assert pts[0] == ov.g1
    for P, h in zip(pts, KAT["ethereum_g1_first5"]):
        assert ov.compress(P).hex() == h and P[0] == int(h, 16) & ((1 << 381) - 1)
    dusk = b"".join(bytes.fromhex(h) for h in KAT["dusk_g1_first5"] + [KAT["dusk_g1_32767"]])
    for P, h in zip(cv.g1_vector_decode(ap_setup.decompress_g1_batch(cv, dusk, gpu)), KAT["dusk_g1_first5"] + [KAT["dusk_g1_32767"]]):
        assert ov.is_on_curve(P) and ov.compress(P).hex() == h


@pytest.mark.parametrize("cname", ["bn254", "bls12-381"])
@pytest.mark.parametrize("log_n", [1, 3, 7])
def test_to_lagrange_matches_known_tau(gpu, cname, log_n):
    """kzg.ToLagrangeG1: out[i] = [L_i(tau)]G1.  With a synthetic SRS the right answer is computable from tau."""
    cv, ov = CURVES[cname]
    n = 1 << log_n
    tau = tau_from_seed(9, cv.r)
    srs = ap_setup.unsafe_srs(cv, max(n, 8), tau, device=gpu)
    pts = srs.g1[: n * 2 * cv.fp_bytes]
    lag = cv.g1_vector_decode(ap_setup.to_lagrange_g1(cv, pts, gpu))
    w = ov.omega(n)
    zn = (pow(tau, n, cv.r) - 1) * pow(n, -1, cv.r) % cv.r
    for i in (0, 1, n // 2, n - 1):
        wi = pow(w, i, cv.r)
        assert lag[i] == ov.mul(ov.g1, wi * zn % cv.r * pow(tau - wi, -1, cv.r) % cv.r)


def test_real_ethereum_srs_2p14_matches_c_oracle(gpu):
    """BASELINE.json configs[2]: BLS12-381 random circuit, 2^14 constraints, Ethereum KZG ceremony SRS (the first
    2^14+3 points of the reference's pk.bin, tests/golden/setup/).  Nobody knows tau, so the checks are:
    (1) byte-identical proof to the C oracle run on the same decompressed SRS; (2) the Lagrange SRS produced by the
    GPU ToLagrangeG1 commits an evaluation vector to the same point as the canonical SRS commits its coefficients."""
    cv, ov = CURVES["bls12-381"]
    info, ok = ap_setup.Get(ap_setup.Name.EthereumKzgCeremonyBLS12381)
    assert ok and info.Trusted
    wl = workloads.random_circuit(cv, 14, 0xA191)
    n = wl.ccs.domain_size()
    srs = ap_setup.trusted_srs(info, n, device=gpu, lagrange=True, root=os.path.join(G, "setup"))
    assert srs.tau is None and len(srs.g1) == (n + 3) * 96 and len(srs.g1_lagrange) == n * 96
    assert cv.g1_from_bytes(srs.g1[:96]) == ov.g1
    pk, vk = ap_plonk.Setup(wl.ccs, srs, device=gpu)
    proof = ap_plonk.Prove(wl.ccs, pk, wl.witness, wl.blinding)
    blob = MarshalProof(proof)
    # (1) C oracle on the same SRS
    clib = c_oracle.load()
    tr = frontend.build_trace(wl.ccs)
    L, R, O = frontend.wire_columns(wl.ccs, wl.solution)
    rc, cblob, _ = c_oracle.prove(clib, cv.abi, n, wl.ccs.GetNbPublicVariables(), srs.g1,
                                  [cv.fr_vector(x) for x in (tr.ql, tr.qr, tr.qm, tr.qo, tr.qk)], tr.perm, cv.fr_vector(L),
                                  cv.fr_vector(R), cv.fr_vector(O), cv.fr_vector(wl.witness.public), cv.fr_vector(wl.blinding),
                                  threads=oracle_threads())
    assert rc == 0 and blob == cblob
    # (1b) the transcribed verifier accepts it with the REAL pairing check against the G2 points of the ceremony's
    # vk.bin (templateLogicSigBLS12_381.go:366-371) - nobody knows tau here
    from oracle import pairing_bls12381 as pr
    from helpers import oracle_vk_from_product
    import dataclasses
    vkb = open(os.path.join(G, "setup", "EethereumKzgCeremonyBLS12_381", "vk.bin"), "rb").read()
    ovk = dataclasses.replace(oracle_vk_from_product(ov, vk), tau=None, g2=(pr.g2_decompress(vkb[:96]), pr.g2_decompress(vkb[96:192])))
    assert ov.decompress(vkb[192:]) == vk.KzgG1                      # Vk.G1 == G1[0] (setup/trusted_setup_test.go:127-129)
    pib = MarshalPublicInputs(wl.witness)
    assert oplonk.verify(ovk, blob, pib)
    bad = bytearray(pib); bad[-1] ^= 1
    assert not oplonk.verify(ovk, blob, bytes(bad))
    # (2) Lagrange vs canonical commitment of the same polynomial, on a context that holds both SRS
    g = SplitMix64(2)
    coeffs = [g.fr(cv.r) for _ in range(n)]
    evals = pk.ntt(coeffs)
    pk.close()
    d = _lib.CircuitDesc()
    # a context with the Lagrange SRS needs a circuit with a commitment; reuse the MSM-only context instead
    ctx = C.c_void_p()
    check(lib.apk_msm_ctx_create(cv.abi, gpu, srs.g1_lagrange, n, 0, C.byref(ctx)))
    out = C.create_string_buffer(96)
    check(lib.apk_msm_g1(ctx, 0, cv.fr_vector(evals), n, out))
    lib.apk_ctx_destroy(ctx)
    ctx2 = C.c_void_p()
    check(lib.apk_msm_ctx_create(cv.abi, gpu, srs.g1, n + 3, 0, C.byref(ctx2)))
    out2 = C.create_string_buffer(96)
    check(lib.apk_msm_g1(ctx2, 0, cv.fr_vector(coeffs), n, out2))
    lib.apk_ctx_destroy(ctx2)
    assert out.raw == out2.raw and any(out.raw)


def test_decompress_rejects_what_gnark_rejects(gpu):
    """kzg SRS ReadFrom -> G1Affine.SetBytes [UPSTREAM] also checks subgroup membership and a clean infinity encoding
    (ADVICE r01): a curve point outside G1 (BLS12-381 has a cofactor), an infinity flag with a payload, x >= p and a
    non-residue x are all refused; the ceremony's own points and a clean infinity are accepted."""
    cv, ov = CURVES["bls12-381"]
    p = cv.p
    x = 5
    while True:                                   # a point of E(Fp) that is NOT in the order-r subgroup
        rhs = (x * x * x + 4) % p
        y = pow(rhs, (p + 1) // 4, p)
        # [r]P as [r-1]P + P: the oracle's mul reduces its scalar mod r
        if y * y % p == rhs and ov.add(ov.mul((x, y), cv.r - 1), (x, y)) is not None:
            break
        x += 1
    from algoplonk_amd import serialize as ser
    good = ser.compress_g1(cv, ov.mul(ov.g1, 12345))
    off_subgroup = ser.compress_g1(cv, (x, y))
    inf_ok = ser.compress_g1(cv, None)
    inf_payload = bytes([0xC0]) + bytes(46) + b"\x01"
    too_big = bytes([0x9F]) + b"\xff" * 47                              # x >= p
    x2 = 5
    while pow((x2 ** 3 + 4) % p, (p - 1) // 2, p) == 1:
        x2 += 1
    not_on_curve = bytes([0x80 | (x2.to_bytes(48, "big")[0])]) + x2.to_bytes(48, "big")[1:]
    out = C.create_string_buffer(2 * 96)
    check(lib.apk_g1_decompress(cv.abi, gpu, good + inf_ok, 2, out))
    assert cv.g1_vector_decode(out.raw) == [ov.mul(ov.g1, 12345), None]
    for bad in (off_subgroup, inf_payload, too_big, not_on_curve):
        assert lib.apk_g1_decompress(cv.abi, gpu, good + bad, 2, out) == _lib.APK_ERR_ARG, bad.hex()
    # BN254 has cofactor 1: every curve point is in G1; flags 00 are not a compressed encoding
    cb, ob = CURVES["bn254"]
    okb = ser.compress_g1(cb, ob.mul(ob.g1, 777))
    out = C.create_string_buffer(64)
    check(lib.apk_g1_decompress(cb.abi, gpu, okb, 1, out))
    assert cb.g1_from_bytes(out.raw) == ob.mul(ob.g1, 777)
    assert lib.apk_g1_decompress(cb.abi, gpu, bytes([okb[0] & 0x3F]) + okb[1:], 1, out) == _lib.APK_ERR_ARG
