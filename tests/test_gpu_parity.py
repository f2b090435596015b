"""-m gpu: the HIP path against the oracle on the same seeded inputs, through the C-ABI (bit-exact)."""
import ctypes as C

import pytest

from algoplonk_amd import _lib, ecc, frontend, plonk as ap_plonk, setup as ap_setup
from algoplonk_amd import MarshalProof, MarshalPublicInputs
from algoplonk_amd._lib import lib, check
from oracle import plonk as oplonk
from oracle.prng import SplitMix64, tau_from_seed

from helpers import CURVES, blinding, oracle_circuit_from_ccs, oracle_vk_from_product, random_chain_ccs, oracle_threads

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


def _setup_pair(cv, ov, ccs, seed, gpu, lagrange=False, msm_window=0, slots=1):
    n = ccs.domain_size()
    tau = tau_from_seed(seed, cv.r)
    srs = ap_setup.unsafe_srs(cv, n, tau, device=gpu, lagrange=lagrange)
    pk, vk = ap_plonk.Setup(ccs, srs, device=gpu, msm_window=msm_window, slots=slots)
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
def test_a_lone_proof_on_a_serving_context_matches_the_oracle(gpu, cname):
    """A proof that has a context of several slots to itself: the commitment of the linearised polynomial is put together on the
    host from GLV halves and the context's fixed-base tables of the VK points (host_msm.h, glv_params.h), the [H] part before the
    evaluations are back (backend_impl.h round 4), on BLS12-381 with the context's parked host threads (the path counter says so).
    Same bytes as the oracle prover's."""
    cv, ov = CURVES[cname]
    ccs, w, sol = random_chain_ccs(cv, 10, 0xA5A5)
    pk, vk, opk, srs = _setup_pair(cv, ov, ccs, 33, gpu, slots=4)
    bl = blinding(cv, 7)
    pk.paths(reset=True)
    blobs = [MarshalProof(ap_plonk.Prove(ccs, pk, w, bl)) for _ in range(3)]
    paths = pk.paths()
    if "APK_HOST_LINCOMB_THREADS" not in os.environ:
        assert paths["proofs"] == 3 and paths["host_lincomb_pooled"] == (0 if cname == "bn254" else 3), paths
    L, R, O = oplonk.solve_lro(oracle_circuit_from_ccs(ov, ccs), sol)
    want = oplonk.marshal_proof(ov, oplonk.prove(opk, L, R, O, w.public, bl))
    assert blobs[0] == want and blobs[1] == want and blobs[2] == want
    pk.close()


@pytest.mark.parametrize("cname", ["bn254", "bls12-381"])
def test_selector_commitments_at_infinity(gpu, cname):
    """A circuit of additions only: Qm = 0 and Qk = 0, so [Qm] and [Qk] in the verifying key are the point at infinity.  They
    still enter every proof's [lin] (coefficients l(zeta) r(zeta) and 1): the host's fixed-base table of an infinity point and the
    plain addition of one must add nothing.  Bytes = the oracle prover's; the transcribed verifier accepts."""
    cv, ov = CURVES[cname]
    r = cv.r
    g = SplitMix64(0x1F)
    nb_public, n = 2, 1 << 6
    sol = [g.fr(r) for _ in range(nb_public + 2)]
    cons = []
    for _ in range(n - nb_public):
        nv = len(sol)
        xa, xb = g.below(nv), g.below(nv)
        ql, qr = g.fr(r), g.fr(r)
        cons.append((ql, qr, 0, r - 1, 0, xa, xb, nv))
        sol.append((ql * sol[xa] + qr * sol[xb]) % r)
    ccs = frontend.ConstraintSystem(r, ["p0", "p1"], ["s0", "s1"], cons, "gates", len(sol))
    w = frontend.Witness(r, sol[:nb_public], sol[nb_public:nb_public + 2])
    pk, vk, opk, srs = _setup_pair(cv, ov, ccs, 5, gpu, slots=3)
    ovk = oracle_vk_from_product(ov, vk)
    assert ovk.qm is None and ovk.qk is None, "the verifying key holds [Qm] = [Qk] = infinity"
    bl = blinding(cv, 12)
    blob = MarshalProof(ap_plonk.Prove(ccs, pk, w, bl))
    L, R, O = oplonk.solve_lro(oracle_circuit_from_ccs(ov, ccs), sol)
    assert blob == oplonk.marshal_proof(ov, oplonk.prove(opk, L, R, O, w.public, bl))
    assert oplonk.verify(ovk, blob, MarshalPublicInputs(w))
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


import json
import os
import threading

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "proof_vectors.json")))["vectors"]


@pytest.mark.parametrize("vec", GOLD, ids=lambda v: "%s-2^%d" % (v["curve"], v["log_n"]))
def test_hip_prover_reproduces_golden_vectors(gpu, vec):
    """tests/golden/proof_vectors.json was produced by oracle/plonk.py in the build container."""
    cv, ov = CURVES[vec["curve"]]
    ccs, w, sol = random_chain_ccs(cv, vec["log_n"], vec["circuit_seed"])
    srs = ap_setup.unsafe_srs(cv, ccs.domain_size(), tau_from_seed(vec["tau_seed"], cv.r), device=gpu)
    pk, vk = ap_plonk.Setup(ccs, srs, device=gpu)
    proof = ap_plonk.Prove(ccs, pk, w, blinding(cv, vec["blinding_seed"]))
    assert MarshalProof(proof).hex() == vec["proof"]
    assert MarshalPublicInputs(w).hex() == vec["public_inputs"]
    assert {k: hex(v) for k, v in proof.challenges.items()} == vec["challenges"]
    assert ov.raw_bytes(vk.Ql).hex() == vec["vk"]["ql"] and ov.raw_bytes(vk.S[2]).hex() == vec["vk"]["s3"]
    pk.close()


@pytest.mark.parametrize("cname", ["bn254", "bls12-381"])
def test_unsatisfied_witness_is_an_error_not_a_proof(gpu, cname, monkeypatch):
    cv, ov = CURVES[cname]
    ccs, w, sol = random_chain_ccs(cv, 6, 5)
    monkeypatch.setenv("APK_WIRES_LAGRANGE", "0")     # no Lagrange SRS given and none derived: basis 1 must be refused below
    pk, vk, opk, srs = _setup_pair(cv, ov, ccs, 3, gpu)
    monkeypatch.delenv("APK_WIRES_LAGRANGE")
    L, R, O = frontend.wire_columns(ccs, sol)
    O[9] = (O[9] + 1) % cv.r
    out = _lib.Proof()
    rc = lib.apk_prove(pk.ctx, cv.fr_vector(L), cv.fr_vector(R), cv.fr_vector(O), cv.fr_vector(w.public),
                       cv.fr_vector(blinding(cv, 1)), None, C.byref(out))
    assert rc == _lib.APK_ERR_WITNESS and b"witness" in lib.apk_last_error()
    # argument errors are reported, never crash
    assert lib.apk_prove(pk.ctx, None, None, None, None, None, None, C.byref(out)) == _lib.APK_ERR_ARG
    with pytest.raises(_lib.ApkError):
        pk.msm([1] * (ccs.domain_size() + 4))          # more scalars than SRS points
    with pytest.raises(_lib.ApkError):
        pk.msm([1, 2, 3], basis=1)                     # no Lagrange SRS in this context
    pk.close()


@pytest.mark.parametrize("cname,window", [("bn254", 7), ("bn254", 9), ("bn254", 12), ("bn254", 16), ("bn254", 17), ("bn254", 18), ("bn254", 20),
                                          ("bls12-381", 8), ("bls12-381", 13), ("bls12-381", 17), ("bls12-381", 19), ("bls12-381", 20)])
def test_msm_window_sizes_and_skewed_scalars(gpu, cname, window):
    """Every window width gives the same group element; skewed inputs (all ones, two distinct values, tiny values)
    stress the bucket work-unit split (full units, sorted remainder units, heavy buckets)."""
    cv, ov = CURVES[cname]
    ccs, w, sol = random_chain_ccs(cv, 9, 77)
    pk, vk, opk, srs = _setup_pair(cv, ov, ccs, 4, gpu, msm_window=window)
    n = ccs.domain_size()
    g = SplitMix64(9)
    cases = [[g.fr(cv.r) for _ in range(n + 3)], [1] * (n + 3), [cv.r - 1] * n, [(i % 2) * 7 + 1 for i in range(n)],
             [g.below(1 << 16) for _ in range(n)], [0] * 5 + [g.fr(cv.r)], [1 << 253] * 3]
    for sc in cases:
        assert pk.msm(sc) == ov.mul(ov.g1, oplonk.poly_eval(sc, srs.tau, cv.r))
    if window >= 16:   # a whole proof through the wide-window sort (c = 17: packed 16-bit LDS counters), batches of 3 included
        bl = blinding(cv, 8)
        oc = oracle_circuit_from_ccs(ov, ccs)
        L, R, O = oplonk.solve_lro(oc, sol)
        assert MarshalProof(ap_plonk.Prove(ccs, pk, w, bl)) == oplonk.marshal_proof(ov, oplonk.prove(opk, L, R, O, w.public, bl))
    pk.close()


def test_concurrent_proofs_are_deterministic(gpu):
    """Several threads share one context (slots): identical inputs -> identical bytes (SURVEY.md §5 determinism)."""
    cv, ov = CURVES["bn254"]
    ccs, w, sol = random_chain_ccs(cv, 10, 31)
    n = ccs.domain_size()
    srs = ap_setup.unsafe_srs(cv, n, tau_from_seed(8, cv.r), device=gpu)
    pk, vk = ap_plonk.Setup(ccs, srs, device=gpu, slots=3)
    bl = blinding(cv, 5)
    blobs = [None] * 6

    def run(i):
        blobs[i] = MarshalProof(ap_plonk.Prove(ccs, pk, w, bl))

    ts = [threading.Thread(target=run, args=(i,)) for i in range(6)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert len(set(blobs)) == 1 and blobs[0] is not None
    assert oplonk.verify(oracle_vk_from_product(ov, vk), blobs[0], MarshalPublicInputs(w))
    # different blinding -> different proof, still accepted
    other = MarshalProof(ap_plonk.Prove(ccs, pk, w, blinding(cv, 6)))
    assert other != blobs[0] and oplonk.verify(oracle_vk_from_product(ov, vk), other, MarshalPublicInputs(w))
    pk.close()


@pytest.mark.parametrize("cname,log_n", [("bn254", 17), ("bls12-381", 14)])
def test_full_size_proof_is_accepted_by_the_transcribed_verifier(gpu, cname, log_n):
    """BASELINE.json configs[1] / configs[2] sizes.  The oracle prover is too slow here; the size-independent property
    is the verifier's acceptance (both KZG openings + the quotient identity at zeta) and rejection under mutation."""
    from algoplonk_amd import workloads
    cv, ov = CURVES[cname]
    wl = workloads.random_circuit(cv, log_n, 0xA190 if cname == "bn254" else 0xA191)
    n = wl.ccs.domain_size()
    srs = ap_setup.unsafe_srs(cv, n, wl.tau, device=gpu)
    pk, vk = ap_plonk.Setup(wl.ccs, srs, device=gpu)
    proof = ap_plonk.Prove(wl.ccs, pk, wl.witness, wl.blinding)
    blob, pib = MarshalProof(proof), MarshalPublicInputs(wl.witness)
    ovk = oracle_vk_from_product(ov, vk)
    assert oplonk.verify(ovk, blob, pib)
    bad = bytearray(blob); bad[400] ^= 1
    assert not oplonk.verify(ovk, bytes(bad), pib)
    # and with the template's real last line: ec.pairing_check against G2 = ([1]G2, [tau]G2)
    import dataclasses
    if cname == "bn254":
        from oracle import pairing_bn254 as pr
    else:
        from oracle import pairing_bls12381 as pr
    ovk_pairing = dataclasses.replace(ovk, tau=None, g2=(pr.G2_GEN, pr.g2_mul(pr.G2_GEN, wl.tau)))
    assert oplonk.verify(ovk_pairing, blob, pib)
    assert not oplonk.verify(ovk_pairing, bytes(bad), pib)
    # byte for byte against the oracle's plain-C prover on the same inputs (BN254 2^17 = the headline config)
    from oracle import c_oracle
    tr = frontend.build_trace(wl.ccs)
    L, R, O = frontend.wire_columns(wl.ccs, wl.solution)
    rc, cblob, cch = c_oracle.prove(c_oracle.load(), cv.abi, n, wl.ccs.GetNbPublicVariables(), srs.g1,
                                    [cv.fr_vector(x) for x in (tr.ql, tr.qr, tr.qm, tr.qo, tr.qk)], tr.perm, cv.fr_vector(L),
                                    cv.fr_vector(R), cv.fr_vector(O), cv.fr_vector(wl.witness.public), cv.fr_vector(wl.blinding),
                                    threads=oracle_threads())
    assert rc == 0 and blob == cblob
    ch = proof.challenges
    assert [ch[k] for k in ("gamma", "beta", "alpha", "zeta", "gamma_kzg")] == cch
    # MSM linearity at full size: msm(a) + msm(b) == msm(a + b)
    a = wl.solution[: n]; b = wl.solution[1: n + 1]
    assert ov.add(pk.msm(a), pk.msm(b)) == pk.msm([(x + y) % cv.r for x, y in zip(a, b)])
    # NTT round trip at full size
    assert pk.ntt(pk.ntt(a), inverse=True) == a
    # the primitives at full size, byte for byte against the C oracle's (orc_msm: classic per-window Pippenger on Jacobian
    # points; orc_ntt: plain radix-2) - SURVEY.md section 8a rows a4 / a6 at BASELINE.json's sizes, not only as properties
    clib = c_oracle.load()
    g = SplitMix64(0xA192)
    sc = cv.fr_vector([g.fr(cv.r) for _ in range(n + 3)])
    want, got = C.create_string_buffer(2 * cv.fp_bytes), C.create_string_buffer(2 * cv.fp_bytes)
    for length in (n + 3, n + 2, n):
        assert clib.orc_msm(cv.abi, srs.g1, sc, length, oracle_threads(), want) == 0
        check(lib.apk_msm_g1(pk.ctx, 0, sc, length, got))
        assert got.raw == want.raw, ("msm", length)
    for which, size in ((0, n), (1, 4 * n)):
        for inverse, coset in ((0, 0), (1, 0)) + (((0, 1), (1, 1)) if which else ()):
            data = sc[: 32 * n] * (size // n)
            a_buf, b_buf = C.create_string_buffer(data, len(data)), C.create_string_buffer(data, len(data))
            assert clib.orc_ntt(cv.abi, a_buf, size, inverse, coset) == 0
            check(lib.apk_ntt(pk.ctx, which, inverse, coset, b_buf))
            assert a_buf.raw == b_buf.raw, ("ntt", which, inverse, coset)
    pk.close()


@pytest.mark.parametrize("cname", ["bn254", "bls12-381"])
def test_msm_only_context_and_index_range_sharding(gpu, cname):
    """BASELINE.json configs[3] on one GPU: four "ranks" each own an index range of the SRS in their own MSM-only
    context; the partial sums added with the library's host curve code equal the unsharded MSM."""
    from algoplonk_amd import parallel
    cv, ov = CURVES[cname]
    n = 1 << 10
    tau = tau_from_seed(17, cv.r)
    srs = ap_setup.unsafe_srs(cv, n, tau, device=gpu)
    g = SplitMix64(0xA192)
    scalars = [g.fr(cv.r) for _ in range(n + 3)]
    sc_bytes = cv.fr_vector(scalars)
    acc = bytes(2 * cv.fp_bytes)
    world = 4
    for rank in range(world):
        sm = parallel.ShardedMsm(cv, srs.g1, device=gpu, comm=parallel.Comm(0, 1), share=(rank, world))
        sm.upload(sc_bytes)
        acc = parallel.g1_add(cv, acc, sm.run())
        # an MSM-only context refuses what it cannot do, loudly
        assert lib.apk_ntt(sm._ctx, 0, 0, 0, sc_bytes) == _lib.APK_ERR_STATE
        sm.close()
    assert cv.g1_from_bytes(acc) == ov.mul(ov.g1, oplonk.poly_eval(scalars, tau, cv.r))


def test_config_3_as_stated_eight_index_ranges_of_2p14(gpu):
    """BASELINE.json configs[3] exactly as written - "BN254 2^17 constraints, single MSM sharded across 8" - on the one GPU of this
    tier: seed 0xA192 (uniform scalars, SRS-shaped points [tau^i]G1, tau = SHA-256(seed) mod r), EIGHT MSM-only
    contexts over the eight 2^14-point index ranges (what the eight ranks of bench.py --mode msm-sharded each hold), the eight
    partial sums added with apk_g1_sum (the local half of the one exchange step) - and the result must be the C oracle's orc_msm
    over all 2^17 pairs, byte for byte."""
    from algoplonk_amd import parallel
    from oracle import c_oracle
    cv, ov = CURVES["bn254"]
    n, world = 1 << 17, 8
    g = SplitMix64(0xA192)
    tau = tau_from_seed(0xA192, cv.r)
    srs = ap_setup.unsafe_srs(cv, n, tau, device=gpu)
    nb = 2 * cv.fp_bytes
    bases = srs.g1[: n * nb]
    sc_bytes = cv.fr_vector([g.fr(cv.r) for _ in range(n)])
    want = C.create_string_buffer(nb)
    assert c_oracle.load().orc_msm(cv.abi, bases, sc_bytes, n, oracle_threads(), want) == 0
    partial = b""
    for rank in range(world):
        sm = parallel.ShardedMsm(cv, bases, device=gpu, comm=parallel.Comm(0, 1), share=(rank, world))
        assert (sm.lo, sm.hi) == (rank << 14, (rank + 1) << 14)
        sm.upload(sc_bytes)
        partial += sm.run()
        sm.close()
    got = C.create_string_buffer(nb)
    check(lib.apk_g1_sum(cv.abi, partial, world, got))
    assert got.raw == want.raw


@pytest.mark.parametrize("cname", ["bn254", "bls12-381"])
def test_msm_over_degenerate_bases_hits_the_doubling_and_cancellation_paths(gpu, cname):
    """SRS-shaped bases from tau = 1, -1 and a 4th root of unity repeat the same few points (and their negatives), so a
    bucket keeps meeting its own sum: the lazy mixed addition's P == Q (doubling) and P == -Q (infinity) branches, and
    additions into an infinity accumulator, all run on the device.  Expected value: (sum s_i tau^i) * G."""
    from algoplonk_amd import parallel
    cv, ov = CURVES[cname]
    n = 1 << 10
    g = SplitMix64(0xDE6E)
    i4 = pow(ov.omega(4), 1, cv.r)
    for tau in (1, cv.r - 1, i4):
        srs = ap_setup.unsafe_srs(cv, n, tau, device=gpu)
        sm = parallel.ShardedMsm(cv, srs.g1, device=gpu, comm=parallel.Comm(0, 1))
        x = g.fr(cv.r)
        cases = [[g.fr(cv.r) for _ in range(n + 3)], [x] * (n + 3), [x if i % 2 == 0 else cv.r - x for i in range(n + 3)],
                 [g.below(4) for _ in range(n + 3)], [1] * (n + 3)]
        for sc in cases:
            sm.upload(cv.fr_vector(sc))
            got = cv.g1_from_bytes(sm.run())
            assert got == ov.mul(ov.g1, oplonk.poly_eval(sc, tau, cv.r)), (cname, tau == 1, len(sc))
        sm.close()


class _Bsb22Circuit(frontend.Circuit):
    """bsb22_test.go:18-39."""
    X = frontend.Public()
    Y = frontend.Secret()

    def __init__(self, nb=1):
        self.nb = nb

    def define(self, api):
        api.AssertIsEqual(self.X, api.Mul(self.Y, self.Y))
        for _ in range(self.nb):
            cmt = api.Commit(self.Y, self.X)
            api.AssertIsDifferent(cmt, 0)


@pytest.mark.parametrize("cname", ["bn254", "bls12-381"])
@pytest.mark.parametrize("nb", [1, 2])
def test_bsb22_proof_marshalling_and_verification(gpu, cname, nb):
    """bsb22_test.go:46-123 (TestBsb22ProofMarshalling) + the oracle prover on the same inputs, byte for byte."""
    cv, ov = CURVES[cname]
    ccs = frontend.Compile(cv.r, _Bsb22Circuit(nb))
    assert len(ccs.commitments) == nb
    n = ccs.domain_size()
    tau = tau_from_seed(41, cv.r)
    srs = ap_setup.unsafe_srs(cv, n, tau, device=gpu, lagrange=True)
    pk, vk = ap_plonk.Setup(ccs, srs, device=gpu)
    a = _Bsb22Circuit(nb); a.X, a.Y = 9, 3
    w = frontend.NewWitness(a, cv.r)
    g = SplitMix64(77)
    hiding = [(g.fr(cv.r), g.fr(cv.r)) for _ in range(nb)]
    bl = blinding(cv, 12)
    proof = ap_plonk.Prove(ccs, pk, w, bl, hiding=hiding)
    blob, pib = MarshalProof(proof), MarshalPublicInputs(w)
    base_words, pt = (24, 64) if cv is ecc.BN254 else (33, 96)
    assert len(blob) == base_words * 32 + nb * 32 + nb * pt                      # bsb22_test.go:97-101
    for i in range(nb):                                                          # :103-120
        assert blob[(base_words + i) * 32:(base_words + i + 1) * 32] == proof.ClaimedValues[6 + i].to_bytes(32, "big")
        start = (base_words + nb) * 32 + i * pt
        assert blob[start:start + pt] == ov.raw_bytes(proof.Bsb22Commitments[i])
    # oracle prover on the same circuit / hiding / blinding
    from oracle import circuits as ocircuits
    oc_ = oplonk.Circuit(ov, ccs.GetNbPublicVariables(), ccs.nb_variables, list(ccs.constraints),
                         [oplonk.Commitment(list(rows), cidx) for rows, cidx in ccs.commitments])
    osrs = oplonk.synthetic_srs(ov, n, tau, materialize=False)
    opk = oplonk.setup(oc_, osrs)
    wn = ov.omega(n)
    pi2 = []
    sol = frontend.solve(ccs, w, lambda col: oplonk.hash_fr(ov.raw_bytes(osrs.commit(oplonk.intt(col, wn, cv.r))), cv.r), hiding, pi2)
    L, R, O = oplonk.solve_lro(oc_, sol)
    opr = oplonk.prove(opk, L, R, O, w.public, bl, pi2=pi2)
    assert proof.Bsb22Commitments == opr.bsb22_commitments
    assert blob == oplonk.marshal_proof(ov, opr)
    assert vk.Qcp == opk.vk.qcp and vk.CommitmentConstraintIndexes == opk.vk.commitment_constraint_indexes
    ovk = oracle_vk_from_product(ov, vk)
    assert oplonk.verify(ovk, blob, pib)
    bad = bytearray(blob); bad[-1] ^= 1
    assert not oplonk.verify(ovk, bytes(bad), pib)
    pk.close()


class _RandomWithCommit(frontend.Circuit):
    """A 2^log_n random-gate circuit (BASELINE.md §2 generator) whose first secret and a mid-circuit wire are BSB22
    committed and whose commitment feeds later gates - the shape of BASELINE.json configs[4] at test size."""
    P0 = frontend.Public()
    P1 = frontend.Public()
    S0 = frontend.Secret()
    S1 = frontend.Secret()

    def __init__(self, gates=0, seed=1, r=0):
        self.gates, self.seed, self.r = gates, seed, r

    def define(self, api):
        g = SplitMix64(self.seed)
        wires = [self.P0, self.P1, self.S0, self.S1]
        for i in range(self.gates):
            a, b = wires[g.below(len(wires))], wires[g.below(len(wires))]
            wires.append(api.Gate(g.fr(self.r), g.fr(self.r), g.fr(self.r), g.fr(self.r), a, b))
            if i == self.gates // 2:
                cmt = api.Commit(self.S0, wires[-1])
                api.AssertIsDifferent(cmt, 0)
                wires.append(cmt)


@pytest.mark.parametrize("cname", ["bn254", "bls12-381"])
def test_bsb22_inside_a_random_circuit_2p10(gpu, cname):
    cv, ov = CURVES[cname]
    circuit = _RandomWithCommit(gates=1000, seed=0xA193, r=cv.r)
    ccs = frontend.Compile(cv.r, circuit)
    n = ccs.domain_size()
    assert n == 1024 and len(ccs.commitments) == 1
    tau = tau_from_seed(0xA193, cv.r)
    srs = ap_setup.unsafe_srs(cv, n, tau, device=gpu, lagrange=True)
    pk, vk = ap_plonk.Setup(ccs, srs, device=gpu)
    a = _RandomWithCommit(gates=1000, seed=0xA193, r=cv.r)
    a.P0, a.P1, a.S0, a.S1 = 11, 22, 33, 44
    w = frontend.NewWitness(a, cv.r)
    g = SplitMix64(5)
    hiding = [(g.fr(cv.r), g.fr(cv.r))]
    bl = blinding(cv, 3)
    proof = ap_plonk.Prove(ccs, pk, w, bl, hiding=hiding)
    blob, pib = MarshalProof(proof), MarshalPublicInputs(w)
    oc_ = oplonk.Circuit(ov, 2, ccs.nb_variables, list(ccs.constraints), [oplonk.Commitment(list(rows), cidx) for rows, cidx in ccs.commitments])
    osrs = oplonk.synthetic_srs(ov, n, tau, materialize=False)
    opk = oplonk.setup(oc_, osrs)
    wn = ov.omega(n)
    pi2 = []
    sol = frontend.solve(ccs, w, lambda col: oplonk.hash_fr(ov.raw_bytes(osrs.commit(oplonk.intt(col, wn, cv.r))), cv.r), hiding, pi2)
    L, R, O = oplonk.solve_lro(oc_, sol)
    opr = oplonk.prove(opk, L, R, O, w.public, bl, pi2=pi2)
    assert blob == oplonk.marshal_proof(ov, opr)
    assert oplonk.verify(oracle_vk_from_product(ov, vk), blob, pib)
    pk.close()


def test_skewed_scalars_at_full_size_are_correct_and_not_pathological(gpu):
    """A Lagrange-basis commitment of a real witness is full of 0 / 1 / small values: one bucket then holds ~n entries.
    The work-unit split + the heavy-bucket merge keep such an MSM within 2x of the uniform case (scalars resident in HBM,
    median of 7 runs each)."""
    import time
    cv, ov = CURVES["bn254"]
    n = 1 << 17
    tau = tau_from_seed(3, cv.r)
    srs = ap_setup.unsafe_srs(cv, n, tau, device=gpu)
    ctx = C.c_void_p()
    check(lib.apk_msm_ctx_create(cv.abi, gpu, srs.g1, n + 3, 0, C.byref(ctx)))
    out = C.create_string_buffer(64)
    g = SplitMix64(1)
    uniform = [g.fr(cv.r) for _ in range(n)]
    cases = {"uniform": uniform, "ones": [1] * n, "bits": [i & 1 for i in range(n)], "small": [g.below(1 << 10) for _ in range(n)],
             "minus_one": [cv.r - 1] * n, "zeros_and_uniform": [0 if i % 8 else uniform[i] for i in range(n)],
             "bytes": [g.below(256) for _ in range(n)]}
    d = C.c_void_p()
    check(lib.apk_device_alloc(ctx, 32 * n, C.byref(d)))
    times = {}
    for name, sc in cases.items():
        buf = cv.fr_vector(sc)
        check(lib.apk_device_upload(ctx, d, buf, len(buf)))
        check(lib.apk_msm_g1_device(ctx, 0, d, n, out))      # warm
        assert cv.g1_from_bytes(out.raw) == ov.mul(ov.g1, oplonk.poly_eval(sc, tau, cv.r)), name
        runs = []
        for _ in range(7):
            t0 = time.perf_counter()
            check(lib.apk_msm_g1_device(ctx, 0, d, n, out))
            runs.append(time.perf_counter() - t0)
        times[name] = sorted(runs)[3]
    check(lib.apk_device_free(ctx, d))
    lib.apk_ctx_destroy(ctx)
    print("skewed MSM times (ms):", {k: round(v * 1e3, 3) for k, v in times.items()})
    # a guard against a pathological path (a serialised heavy bucket was 9x uniform once), not a performance figure: wall-clock
    # ratios on a shared box do not belong in the parity tier, so the bound is far from the measured 1.0 - 1.8x
    assert max(times.values()) < 6.0 * times["uniform"] + 0.001, times


@pytest.mark.parametrize("cname,log_n", [("bls12-381", 17), ("bn254", 16)])
def test_bsb22_at_scale_is_accepted_by_the_transcribed_verifier(gpu, cname, log_n):
    """SURVEY.md §8f.2 / BASELINE.json configs[4] shape on one GPU: a large random circuit with one BSB22 commitment over
    16 wires (sparse committed column, Lagrange-SRS MSM in the hint).  Verified by the template transcription, including
    the hash_fr / L_{nbPublic+cci}(zeta) public-input term (templateLogicSigBN254.go:187-193)."""
    from algoplonk_amd import workloads
    cv, ov = CURVES[cname]
    ccs, w, bl, tau = workloads.random_circuit_bsb22(cv, log_n, 0xA193)
    n = ccs.domain_size()
    assert n == 1 << log_n and len(ccs.commitments) == 1
    srs = ap_setup.unsafe_srs(cv, n, tau, device=gpu, lagrange=True)
    pk, vk = ap_plonk.Setup(ccs, srs, device=gpu)
    proof = ap_plonk.Prove(ccs, pk, w, bl, hiding=[(12345, 67890)])
    blob, pib = MarshalProof(proof), MarshalPublicInputs(w)
    base_words, pt = (24, 64) if cv is ecc.BN254 else (33, 96)
    assert len(blob) == base_words * 32 + 32 + pt
    ovk = oracle_vk_from_product(ov, vk)
    assert oplonk.verify(ovk, blob, pib)
    bad = bytearray(blob); bad[-1] ^= 1           # tamper with the BSB22 commitment point
    assert not oplonk.verify(ovk, bytes(bad), pib)
    bad = bytearray(blob); bad[base_words * 32 + 5] ^= 1   # tamper with qcp(zeta)
    assert not oplonk.verify(ovk, bytes(bad), pib)
    # byte for byte against the host prover's BSB22 path (oracle/fast_prover.c, itself held to oracle/plonk.py on the reference's
    # commitment circuit by tests/test_oracle_c.py): the same solved wires, committed column, blinding
    from oracle import c_oracle
    solution, pi2 = ap_plonk.solve_with_commitments(ccs, pk, w, hiding=[(12345, 67890)])
    tr = frontend.build_trace(ccs)
    L, R, O = frontend.wire_columns(ccs, solution)
    fp = c_oracle.FastProver(c_oracle.load(), cv.abi, n, ccs.GetNbPublicVariables(), srs.g1,
                             [cv.fr_vector(x) for x in (tr.ql, tr.qr, tr.qm, tr.qo, tr.qk)], tr.perm, threads=oracle_threads(),
                             qcp=[cv.fr_vector(q) for q in tr.qcp], cci=[cidx for _, cidx in ccs.commitments], srs_lagrange=srs.g1_lagrange)
    rc, want, _ = fp.prove(cv.fr_vector(L), cv.fr_vector(R), cv.fr_vector(O), cv.fr_vector(w.public), cv.fr_vector(bl),
                           threads=oracle_threads(), pi2=[cv.fr_vector(p) for p in pi2])
    fp.close()
    assert rc == 0 and blob == want
    pk.close()


def test_largest_size_bls12_381_2p21_with_bsb22_proof_verifies(gpu):
    """BASELINE.json configs[4] AS STATED, on one GPU: BLS12-381, n = 2^21 (64 MiB per polynomial, 2 x 3.2 GB of windowed
    tables: canonical + Lagrange SRS) WITH one BSB22 commitment (committed column: 16 wires + 2 hiding entries out of 2^21).
    The proof is checked by the transcribed verifier, whose cost does not depend on n, including the hash_fr /
    L_{nbPublic+cci}(zeta) public-input term (templateLogicSigBLS12_381.go twin of templateLogicSigBN254.go:187-193)."""
    from algoplonk_amd import workloads
    cv, ov = CURVES["bls12-381"]
    ccs, w, bl, tau = workloads.random_circuit_bsb22(cv, 21, 0xA193)
    n = ccs.domain_size()
    assert n == 1 << 21 and len(ccs.commitments) == 1
    srs = ap_setup.unsafe_srs(cv, n, tau, device=gpu, lagrange=True)
    pk, vk = ap_plonk.Setup(ccs, srs, device=gpu)
    proof = ap_plonk.Prove(ccs, pk, w, bl, hiding=[(0xA193, 0x3910A)])
    blob, pib = MarshalProof(proof), MarshalPublicInputs(w)
    ovk = oracle_vk_from_product(ov, vk)
    assert len(blob) == 1056 + 32 + 96 and oplonk.verify(ovk, blob, pib)     # bsb22_test.go:97-101
    assert proof.Bsb22Commitments[0] is not None
    bad = bytearray(blob); bad[900] ^= 1
    assert not oplonk.verify(ovk, bytes(bad), pib)
    bad = bytearray(blob); bad[-1] ^= 1            # the BSB22 commitment point
    assert not oplonk.verify(ovk, bytes(bad), pib)
    bad = bytearray(pib); bad[-1] ^= 1             # testutils/verifier_integration_test.go:188-228: flipped public input
    assert not oplonk.verify(ovk, blob, bytes(bad))
    # ONE MSM at this size byte for byte against the C oracle (orc_msm: per-window Pippenger on Jacobian points, all host
    # threads): configs[4] is not held by properties alone.  n + 3 pairs = the whole canonical SRS, and n pairs.
    from oracle import c_oracle
    clib = c_oracle.load()
    import hashlib
    raw = bytearray(hashlib.shake_256(b"apk configs[4] msm 0xA193").digest(32 * (n + 3)))      # seeded, 64 MiB in a fraction of a second
    raw[31::32] = bytes(b & 0x1F for b in raw[31::32])      # little-endian limbs: below 2^253 < r, a valid Montgomery-form scalar
    sc = bytes(raw)
    want, got = C.create_string_buffer(2 * cv.fp_bytes), C.create_string_buffer(2 * cv.fp_bytes)
    for length in (n + 3, n):
        assert clib.orc_msm(cv.abi, srs.g1, sc, length, oracle_threads(), want) == 0
        check(lib.apk_msm_g1(pk.ctx, 0, sc, length, got))
        assert got.raw == want.raw, ("msm at 2^21", length)
    pk.close()


def test_public_api_pythagorean_compile_verify_export(gpu, tmp_path):
    """BASELINE.json configs[0] through the HIP path and the PUBLIC API: the reference's examples/basic workload
    (/root/reference/examples/basic/logicsigVerifier/main.go:30-52: a^2 + b^2 == c^2, assignment 3, 4, 5) run as
    Compile (algoplonk.go:37-59) -> CompiledCircuit.Verify (algoplonk.go:79-98: prove + verify) ->
    ExportProofAndPublicInputs (algoplonk.go:103-132), on BN254 with the TestOnly setup.  The
    exported 768-byte blob must equal the oracle prover's for the same tau / blinding, and the 64-byte public-input file
    the oracle's marshalling."""
    from algoplonk_amd import Compile

    class Pyth(frontend.Circuit):
        A = frontend.Public(); B = frontend.Public(); C = frontend.Secret()

        def define(self, api):
            api.AssertIsEqual(api.Add(api.Mul(self.A, self.A), api.Mul(self.B, self.B)), api.Mul(self.C, self.C))

    for cname, name in (("bn254", ap_setup.Name.TestOnlyBN254), ("bls12-381", ap_setup.Name.TestOnlyBLS12381)):
        cv, ov = CURVES[cname]
        seed = 0x5EED0000 + int(name)
        cc = Compile(Pyth(), cv, name, device=gpu, seed=seed)
        assert cc.Vk.Size == 8 and cc.Vk.NbPublicVariables == 2
        a = Pyth(); a.A, a.B, a.C = 3, 4, 5
        bl = blinding(cv, 0xB1)
        seen = []

        def transcribed(vk, blob, pib):
            seen.append((blob, pib))
            return oplonk.verify(oracle_vk_from_product(ov, vk), blob, pib)

        vp = cc.Verify(a, blinding=bl, verifier=transcribed)
        assert len(seen) == 1
        pf, pif = str(tmp_path / ("proof_%s.bin" % cname)), str(tmp_path / ("pi_%s.bin" % cname))
        vp.ExportProofAndPublicInputs(pf, pif)
        blob, pib = open(pf, "rb").read(), open(pif, "rb").read()
        assert (blob, pib) == seen[0]
        assert len(blob) == (768 if cv is ecc.BN254 else 1056) and len(pib) == 64
        assert pib == (3).to_bytes(32, "big") + (4).to_bytes(32, "big")
        # the oracle prover on the same circuit / SRS / witness / blinding
        tau = int.from_bytes(seed.to_bytes(48, "big"), "big") % cv.r
        oc = oracle_circuit_from_ccs(ov, cc.Ccs)
        opk = oplonk.setup(oc, oplonk.synthetic_srs(ov, 8, tau, materialize=False))
        sol = frontend.solve(cc.Ccs, frontend.NewWitness(a, cv.r))
        L, R, O = oplonk.solve_lro(oc, sol)
        want = oplonk.marshal_proof(ov, oplonk.prove(opk, L, R, O, [3, 4], bl))
        assert blob == want
        # the built-in verifier (algoplonk.go:93 runs plonk.Verify unconditionally) accepts it too, and a wrong assignment
        # never yields a VerifiedProof (algoplonk.go:90-92)
        cc.Verify(a, blinding=bl)
        wrong = Pyth(); wrong.A, wrong.B, wrong.C = 3, 4, 6
        with pytest.raises(RuntimeError, match="error creating Plonk proof"):
            cc.Verify(wrong, blinding=bl)
        cc.Pk.close()


@pytest.mark.parametrize("cname", ["bn254", "bls12-381"])
def test_commit_hook_and_ranged_batch_msm(gpu, cname):
    """SURVEY.md section 8e row 2 on ONE GPU: (1) apk_msm_g1_batch_device = partial commitments over index ranges, against
    the oracle; (2) a proof whose commitments all go through the context's commit hook (libapk's communicator at world = 1: the
    same C code path the ranks of a node run, minus the exchange) is byte-identical to the plain proof - with and without BSB22."""
    from algoplonk_amd import parallel, workloads
    cv, ov = CURVES[cname]
    ccs, w, sol = random_chain_ccs(cv, 9, 0xA190 + 9)
    pk, vk, opk, srs = _setup_pair(cv, ov, ccs, 31, gpu)
    n, tau = ccs.domain_size(), srs.tau
    g = SplitMix64(17)
    sc = [g.fr(cv.r) for _ in range(n + 3)]
    buf = cv.fr_vector(sc)
    d = C.c_void_p()
    check(lib.apk_device_alloc(pk.ctx, len(buf), C.byref(d)))
    check(lib.apk_device_upload(pk.ctx, d, buf, len(buf)))
    segs = [(0, 100), (100, n + 3), (37, 38)]
    k = len(segs)
    ptrs, offs, ls = (C.c_void_p * k)(), (C.c_uint64 * k)(), (C.c_uint64 * k)()
    for i, (lo, hi) in enumerate(segs):
        ptrs[i], offs[i], ls[i] = d.value + 32 * lo, lo, hi - lo
    out = C.create_string_buffer(k * 2 * cv.fp_bytes)
    check(lib.apk_msm_g1_batch_device(pk.ctx, 0, k, ptrs, offs, ls, out))
    got = cv.g1_vector_decode(out.raw)
    for (lo, hi), P in zip(segs, got):
        assert P == ov.mul(ov.g1, sum(sc[i] * pow(tau, i, cv.r) for i in range(lo, hi)) % cv.r)
    assert ov.add(got[0], got[1]) == pk.msm(sc)
    offs[1] = n                                                        # range past the SRS: an error, not a wild read
    assert lib.apk_msm_g1_batch_device(pk.ctx, 0, k, ptrs, offs, ls, out) == _lib.APK_ERR_ARG
    check(lib.apk_device_free(pk.ctx, d))
    # (2) the hook path
    bl = blinding(cv, 5)
    plain = MarshalProof(ap_plonk.Prove(ccs, pk, w, bl))
    split = parallel.Comm(0, 1).bind(pk.ctx)
    split.split_begin()
    hooked = MarshalProof(ap_plonk.Prove(ccs, pk, w, bl))
    split.split_end()
    split.close()
    assert hooked == plain and MarshalProof(ap_plonk.Prove(ccs, pk, w, bl)) == plain
    pk.close()
    ccs2, w2, bl2, tau2 = workloads.random_circuit_bsb22(cv, 8, 0xA193, nb_commitments=1, committed=4)
    srs2 = ap_setup.unsafe_srs(cv, ccs2.domain_size(), tau2, device=gpu, lagrange=True)
    pk2, vk2 = ap_plonk.Setup(ccs2, srs2, device=gpu)
    plain2 = MarshalProof(ap_plonk.Prove(ccs2, pk2, w2, bl2, hiding=[(3, 4)]))
    split2 = parallel.Comm(0, 1).bind(pk2.ctx)
    split2.split_begin()
    assert MarshalProof(ap_plonk.Prove(ccs2, pk2, w2, bl2, hiding=[(3, 4)])) == plain2
    split2.split_end()
    split2.close()
    pk2.close()


@pytest.mark.parametrize("cname", ["bn254", "bls12-381"])
@pytest.mark.parametrize("nb_public", [0, 1, 7, 40])
def test_prove_with_few_and_many_public_inputs(gpu, cname, nb_public):
    """Edge cases of the public-input handling on random-gate circuits: none, one (as many as the reference's identity circuit
    has - that circuit itself, with [Qm] at infinity, is proved in tests/test_template_pin.py), more rows
    than the quotient kernel's direct Qk completion takes (falls back to iNTT + coset NTT of the completed column), and
    enough of them that PI(zeta) - the host-side sum behind lin(zeta) - has many Lagrange terms.  Byte-identical to the oracle."""
    cv, ov = CURVES[cname]
    ccs, w, sol = random_chain_ccs(cv, 6, 0xA190 + nb_public, nb_public=nb_public)
    assert ccs.GetNbPublicVariables() == nb_public
    pk, vk, opk, srs = _setup_pair(cv, ov, ccs, 77, gpu)
    bl = blinding(cv, 3)
    proof = ap_plonk.Prove(ccs, pk, w, bl)
    blob, pib = MarshalProof(proof), MarshalPublicInputs(w)
    oc = oracle_circuit_from_ccs(ov, ccs)
    L, R, O = oplonk.solve_lro(oc, sol)
    want = oplonk.marshal_proof(ov, oplonk.prove(opk, L, R, O, w.public, bl))
    assert blob == want and len(pib) == 32 * nb_public
    assert oplonk.verify(oracle_vk_from_product(ov, vk), blob, pib)
    ap_plonk.Verify(proof, vk, w)                                      # the library's own verifier agrees
    if nb_public:
        bad = frontend.Witness(w.field, [(w.public[0] + 1) % cv.r] + w.public[1:], w.secret)
        with pytest.raises(ap_plonk.VerificationError):
            ap_plonk.Verify(proof, vk, bad)
    pk.close()


_VARIANT_SCRIPT = r"""
import hashlib, sys, os
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from algoplonk_amd import MarshalProof, plonk, setup
from oracle.prng import tau_from_seed
from helpers import CURVES, blinding, random_chain_ccs
cname, log_n, window = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
cv, _ = CURVES[cname]
ccs, w, _ = random_chain_ccs(cv, log_n, 77)
srs = setup.unsafe_srs(cv, ccs.domain_size(), tau_from_seed(9, cv.r), device=0)
pk, vk = plonk.Setup(ccs, srs, device=0, msm_window=window, slots=3)
print("SHA", hashlib.sha256(MarshalProof(plonk.Prove(ccs, pk, w, blinding(cv, 3)))).hexdigest())
"""


@pytest.mark.parametrize("cname,log_n,window,variants", [
    ("bn254", 12, 16, [{"APK_MSM_LEAN_TAIL": "1"}, {"APK_MSM_LEAN_TAIL": "0", "APK_TAIL_FILL": "0"}, {"APK_TAIL_FILL": "2"},
                       {"APK_NTT_RADIX4": "1"}, {"APK_NTT_RADIX4": "1", "APK_NTT_THREADS": "64"}, {"APK_MSM_SORT2": "1"},
                       # round 4: the layouts the large sorts take - 2 048 partitions of 16 buckets with wave-private counters,
                       # sixteen lanes per run in the copy-out, the three-launch partition scan - forced at a size the test can afford
                       {"APK_MSM_SORT2": "1", "APK_MSM_PART_PBLOG": "4"}, {"APK_MSM_SORT2": "1", "APK_MSM_PART_PBLOG": "6", "APK_MSM_PART_SMALL_SCAN": "0"},
                       {"APK_MSM_SORTED_MERGE": "0"}, {"APK_MSM_LEAN_TAIL": "1", "APK_MSM_ROWCOL_LANES": "8"},
                       # the four-launch form of the two-level sort (count - scan - scatter - sort) behind the two-launch default
                       {"APK_MSM_SORT2": "1", "APK_MSM_SORT_FUSED": "0"}, {"APK_MSM_SORT2": "1", "APK_MSM_SORT_FUSED": "0", "APK_MSM_PART_PBLOG": "4"},
                       {"APK_MSM_SCAN_FUSED": "1"}, {"APK_MSM_GRAPH": "1", "APK_MSM_SORT2": "1"},
                       {"APK_MSM_COMBINE_QUAD": "1", "APK_MSM_LEAN_TAIL": "0"}, {"APK_MSM_COMBINE_QUAD": "0", "APK_MSM_LEAN_TAIL": "0"},
                       # round 5: the host's [lin] combination - GLV halves, fixed-base tables of the VK points, the [H] part ahead of
                       # the evaluations, parked threads - against the full-length one-pass Straus form it replaced
                       {"APK_HOST_LINCOMB_THREADS": "3"}, {"APK_HOST_GLV": "0", "APK_HOST_FIXED": "0", "APK_LIN_EARLY_H": "0"}, {"APK_ZERO_COPY": "0"},
                       {"APK_HOST_GLV": "0"}, {"APK_HOST_FIXED": "0", "APK_LIN_EARLY_H": "0", "APK_HOST_LINCOMB_THREADS": "4"},
                       # round 6: twiddle tables kept as unsaturated-limb records, alone and with the radix-4 steps; the stream waits' forms
                       {"APK_NTT_TWU": "1"}, {"APK_NTT_TWU": "1", "APK_NTT_RADIX4": "1"}, {"APK_SYNC_BLOCKING": "2"}, {"APK_SYNC_BLOCKING": "1"}]),
    ("bls12-381", 10, 12, [{"APK_MSM_LEAN_TAIL": "1", "APK_NTT_RADIX4": "1"}, {"APK_MSM_SORT2": "1"},
                           {"APK_MSM_SORT2": "1", "APK_MSM_PART_PBLOG": "3", "APK_MSM_PART_SMALL_SCAN": "0"},
                           {"APK_MSM_COMBINE_QUAD": "1", "APK_MSM_LEAN_TAIL": "0"}, {"APK_MSM_COMBINE_QUAD": "0", "APK_MSM_LEAN_TAIL": "0"},
                           {"APK_HOST_LINCOMB_THREADS": "1"}, {"APK_HOST_GLV": "0", "APK_HOST_FIXED": "0", "APK_LIN_EARLY_H": "0"},
                           {"APK_HOST_FIXED": "0"}, {"APK_ZERO_COPY": "0"}, {"APK_NTT_TWU": "1", "APK_NTT_RADIX4": "1"}, {"APK_SYNC_BLOCKING": "2"}]),
])
def test_run_time_variants_give_the_same_bytes(gpu, cname, log_n, window, variants):
    """The forms the library picks at run time - lean tail kernels when other proofs are in flight (sixteen-lane row/column
    sums, one lane per bucket in the merge), the tail-filling side stream of a lone proof, radix-4 NTT steps above 2^19, the
    two-level counting sort that is built but off by default - are
    scheduling choices: every one of them, forced through its environment knob in a process of its own (the knobs are read
    once), must produce the proof bytes of the default build.  c = 16 at BN254 so the merge kernel's lean form (>= 32 k buckets)
    is really taken."""
    import subprocess
    import sys

    def sha(extra):
        env = dict(os.environ)
        env.update(extra)
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
        out = subprocess.run([sys.executable, "-c", _VARIANT_SCRIPT, cname, str(log_n), str(window)], cwd=root, env=env,
                             capture_output=True, text=True, timeout=240)
        assert out.returncode == 0, out.stderr[-2000:]
        return [ln.split()[1] for ln in out.stdout.splitlines() if ln.startswith("SHA ")][0]

    want = sha({})
    for v in variants:
        assert sha(v) == want, v


_SKEWED_SORT2_SCRIPT = r"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from algoplonk_amd import setup
from algoplonk_amd._lib import lib, check
from oracle import plonk as oplonk
from oracle.prng import SplitMix64, tau_from_seed
from helpers import CURVES
cv, ov = CURVES["bn254"]
n = 1 << 16
tau = tau_from_seed(3, cv.r)
srs = setup.unsafe_srs(cv, n, tau, device=0)
ctx = C.c_void_p()
check(lib.apk_msm_ctx_create(cv.abi, 0, srs.g1, n + 3, 0, C.byref(ctx)))
out = C.create_string_buffer(64)
g = SplitMix64(1)
uniform = [g.fr(cv.r) for _ in range(n)]
cases = {"uniform": uniform, "ones": [1] * n, "minus_one": [cv.r - 1] * n, "zeros_and_uniform": [0 if i % 8 else uniform[i] for i in range(n)],
         "two_values": [uniform[i & 1] for i in range(n)]}
d = C.c_void_p()
check(lib.apk_device_alloc(ctx, 32 * n, C.byref(d)))
for name, sc in cases.items():
    buf = cv.fr_vector(sc)
    check(lib.apk_device_upload(ctx, d, buf, len(buf)))
    check(lib.apk_msm_g1_device(ctx, 0, d, n, out))
    assert cv.g1_from_bytes(out.raw) == ov.mul(ov.g1, oplonk.poly_eval(sc, tau, cv.r)), name
print("SKEWED_OK")
"""


@pytest.mark.parametrize("extra", [{}, {"APK_MSM_PART_PBLOG": "4", "APK_MSM_PART_SMALL_SCAN": "0"}, {"APK_MSM_SORT_FUSED": "0"},
                                   {"APK_MSM_SORT_FUSED": "0", "APK_MSM_PART_PBLOG": "4", "APK_MSM_PART_SMALL_SCAN": "0"}],
                         ids=["default-layout", "16-bucket-partitions", "four-launch", "four-launch-16-bucket-partitions"])
def test_two_level_sort_with_skewed_scalars(gpu, extra):
    """The two-level sort's overflow paths: all-equal scalars put a whole MSM's entries into W buckets, so the partitions that
    hold them are far larger than the LDS tile of the second level (it then scatters in HBM) and every other partition is
    empty.  Forced on (it is only picked under load) in a process of its own, with the lean tail forms; checked against the
    known-tau shortcut like the one-level test above."""
    import subprocess
    import sys
    env = dict(os.environ)
    env.update({"APK_MSM_SORT2": "1", "APK_MSM_LEAN_TAIL": "1"})
    env.update(extra)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
    out = subprocess.run([sys.executable, "-c", _SKEWED_SORT2_SCRIPT], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "SKEWED_OK" in out.stdout, (out.stdout[-500:], out.stderr[-2000:])


@pytest.mark.parametrize("cname,log_n", [("bn254", 17), ("bls12-381", 14)])
def test_proofs_under_load_match_the_c_oracle(gpu, cname, log_n):
    """the loaded path at the headline size.  32 callers x 5 rounds on a 32-slot context (the library caps
    it at 16 proving slots; the other callers wait their turn, which is bench.py's configuration) prove the same instance with
    the same blinding; EVERY one of the 160 blobs must be the bytes of the C oracle's proof ("safe to call
    concurrently from several goroutines", algoplonk.go:89).  Under load the library takes other kernels than for a lone proof
    (backend_impl.h: two-level sort below 2^16 bases, lean merge and sixteen-lane row/column sums, radix-4 NTT steps from 2^17):
    the context's path counters must show that those forms produced the blobs compared here - and that a lone proof on the same
    context, which takes the latency forms (tail filling, tree tails), gives the same bytes again."""
    import hashlib
    import threading
    from algoplonk_amd import workloads
    from oracle import c_oracle
    cv, ov = CURVES[cname]
    wl = workloads.random_circuit(cv, log_n, 0xA190 if cname == "bn254" else 0xA191)
    n = wl.ccs.domain_size()
    srs = ap_setup.unsafe_srs(cv, n, wl.tau, device=gpu)
    T, rounds = 32, 5
    pk, vk = ap_plonk.Setup(wl.ccs, srs, device=gpu, slots=T)
    tr = frontend.build_trace(wl.ccs)
    L, R, O = frontend.wire_columns(wl.ccs, wl.solution)
    rc, want, _ = c_oracle.prove(c_oracle.load(), cv.abi, n, wl.ccs.GetNbPublicVariables(), srs.g1,
                                 [cv.fr_vector(x) for x in (tr.ql, tr.qr, tr.qm, tr.qo, tr.qk)], tr.perm, cv.fr_vector(L),
                                 cv.fr_vector(R), cv.fr_vector(O), cv.fr_vector(wl.witness.public), cv.fr_vector(wl.blinding),
                                 threads=oracle_threads())
    assert rc == 0
    dptr = []
    for b in (cv.fr_vector(v) for v in (L, R, O)):
        p = C.c_void_p()
        check(lib.apk_device_alloc(pk.ctx, len(b), C.byref(p)))
        check(lib.apk_device_upload(pk.ctx, p, b, len(b)))
        dptr.append(p)
    pub, bl = cv.fr_vector(wl.witness.public), cv.fr_vector(wl.blinding)
    blobs, errors, lock = {}, [], threading.Lock()

    def worker():
        pr = _lib.Proof()
        out = C.create_string_buffer(2048)
        ln = C.c_size_t(0)
        for _ in range(rounds):
            rc_ = lib.apk_prove_device(pk.ctx, dptr[0], dptr[1], dptr[2], pub, bl, None, C.byref(pr))
            if rc_ != 0:
                errors.append((rc_, lib.apk_last_error()))
                return
            check(lib.apk_marshal_proof(C.byref(pr), out, 2048, C.byref(ln)))
            with lock:
                blobs[out.raw[: ln.value]] = blobs.get(out.raw[: ln.value], 0) + 1

    pk.paths(reset=True)
    th = [threading.Thread(target=worker) for _ in range(T)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errors, errors[0]
    assert sum(blobs.values()) == T * rounds
    assert list(blobs) == [want], "proofs made under load differ from the C oracle's: %s" % sorted(hashlib.sha256(b).hexdigest()[:12] for b in blobs)
    loaded = pk.paths(reset=True)
    assert loaded["proofs"] == T * rounds
    # the loaded forms ran (nearly every batch has other proofs beside it; the first and last few may not)
    assert loaded["msm_lean_tail"] >= loaded["msm_batches"] // 2, loaded
    assert loaded["msm_rowcol_serial"] >= loaded["msm_batches"] // 2, loaded
    if log_n >= 16:
        assert loaded["msm_units_by_load"] >= loaded["msm_batches"] // 2, loaded        # 48-entry accumulate units (from 2^16 bases)
    assert loaded["msm_sort_two_level"] >= loaded["msm_batches"] // 2, loaded
    if log_n >= 17:
        assert loaded["ntt_radix4_by_load"] >= 1, loaded
    else:
        assert loaded["msm_sort_two_level_by_load"] >= loaded["msm_batches"] // 2, loaded      # 2^14 bases: two-level only under load
    # ... and a lone proof on the same context: the latency forms, the same bytes
    lone = MarshalProof(ap_plonk.Proof(cv, _prove_resident(pk, dptr, pub, bl)))
    alone = pk.paths(reset=True)
    assert lone == want
    assert alone["proofs"] == 1 and alone["msm_lean_tail"] == 0 and alone["msm_rowcol_serial"] == 0 and alone["tail_fill_proofs"] == 1 and alone["msm_units_by_load"] == 0, alone
    for p in dptr:
        check(lib.apk_device_free(pk.ctx, p))
    pk.close()


def _prove_resident(pk, dptr, pub, bl):
    pr = _lib.Proof()
    check(lib.apk_prove_device(pk.ctx, dptr[0], dptr[1], dptr[2], pub, bl, None, C.byref(pr)))
    return pr


@pytest.mark.parametrize("cname", ["bn254", "bls12-381"])
def test_lagrange_basis_wire_commitments_are_the_same_group_elements(gpu, cname, monkeypatch):
    """Round 5: [L][R][O] over the Lagrange SRS the way gnark commits them (setup/setup.go:124,138 builds that SRS for it) - one
    MSM over the witness values + the blinding scalars on the points [tau^(n+k)] - [tau^k] - is the same group element as the
    canonical-basis commitment of the blinded polynomial, so every proof stays byte-identical to the oracle's, whichever route a
    context takes.  auto: a bit-heavy witness (workloads.skewed_circuit: ~80 % of the wire values in {0, 1}) takes the Lagrange
    route from its first proof, uniform wires stay canonical; APK_WIRES_LAGRANGE = 1 / 0 force either (read when the context is
    created).  The Lagrange table is derived on the device when the caller gives no Lagrange SRS, and taken from the caller's
    (as gnark's ProvingKey.KzgLagrange) when it does - both are exercised."""
    from algoplonk_amd import workloads
    cv, ov = CURVES[cname]
    log_n = 10
    for kind, mode, given, expect_lag in (("bits", None, False, True), ("bits", None, True, True), ("bits", "0", False, False),
                                          ("uniform", None, False, False), ("uniform", "1", True, True), ("uniform", "1", False, True)):
        wl = (workloads.skewed_circuit if kind == "bits" else workloads.random_circuit)(cv, log_n, 0xB175)
        n = wl.ccs.domain_size()
        if mode is None:
            monkeypatch.delenv("APK_WIRES_LAGRANGE", raising=False)
        else:
            monkeypatch.setenv("APK_WIRES_LAGRANGE", mode)
        srs = ap_setup.unsafe_srs(cv, n, wl.tau, device=gpu, lagrange=given)
        pk, vk = ap_plonk.Setup(wl.ccs, srs, device=gpu, slots=2)
        oc = oracle_circuit_from_ccs(ov, wl.ccs)
        opk = oplonk.setup(oc, oplonk.synthetic_srs(ov, n, wl.tau, materialize=False))
        L, R, O = oplonk.solve_lro(oc, wl.solution)
        want = oplonk.marshal_proof(ov, oplonk.prove(opk, L, R, O, wl.witness.public, wl.blinding))
        pk.paths(reset=True)
        for _ in range(3):
            assert MarshalProof(ap_plonk.Prove(wl.ccs, pk, wl.witness, wl.blinding)) == want, (kind, mode, given)
        got = pk.paths(reset=True)
        assert got["proofs"] == 3 and got["msm_lagrange_wires"] == (3 if expect_lag else 0), (kind, mode, given, got)
        # the extended Lagrange table through the primitive: basis 1, the n Lagrange points followed by the three blinding points
        if given or mode != "0":
            g = SplitMix64(5)
            sc = [g.below(3) for _ in range(n)] + [g.fr(cv.r) for _ in range(3)]
            f_at_tau = oplonk.poly_eval(oplonk.intt(sc[:n], ov.omega(n), cv.r), wl.tau, cv.r)
            blind = sum(b * (pow(wl.tau, n + k, cv.r) - pow(wl.tau, k, cv.r)) for k, b in enumerate(sc[n:])) % cv.r
            assert pk.msm(sc, basis=1) == ov.mul(ov.g1, (f_at_tau + blind) % cv.r)
        pk.close()


@pytest.mark.parametrize("cname", ["bn254", "bls12-381"])
@pytest.mark.parametrize("log_n", [3, 5])
def test_lagrange_route_at_the_smallest_sizes(gpu, cname, log_n, monkeypatch):
    """The extended Lagrange table at n = 8 and n = 32 (n + 3 = 11 / 35 bases, the blinding points right behind a handful of Lagrange
    points), forced: the proofs are the oracle's byte for byte, with and without public inputs in the witness rows."""
    cv, ov = CURVES[cname]
    monkeypatch.setenv("APK_WIRES_LAGRANGE", "1")
    for nb_public in (1, 2):
        ccs, w, sol = random_chain_ccs(cv, log_n, 90 + log_n, nb_public=nb_public)
        pk, vk, opk, srs = _setup_pair(cv, ov, ccs, 6, gpu)
        bl = blinding(cv, 15)
        oc = oracle_circuit_from_ccs(ov, ccs)
        L, R, O = oplonk.solve_lro(oc, sol)
        want = oplonk.marshal_proof(ov, oplonk.prove(opk, L, R, O, w.public, bl))
        for _ in range(2):
            assert MarshalProof(ap_plonk.Prove(ccs, pk, w, bl)) == want
        assert pk.paths()["msm_lagrange_wires"] == 2
        pk.close()
