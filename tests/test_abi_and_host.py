"""CPU tier: the C-ABI library loads, exports every symbol include/apk.h declares, and its host-only entry points
(wire formats, conversions, the host instantiation of the device arithmetic templates) agree with the oracle.
No compute entry point is called here: without a GPU they must fail loudly, which is asserted."""
import ctypes as C
import os
import random
import re
import subprocess

import pytest

from algoplonk_amd import _lib, ecc, frontend, setup as ap_setup
from algoplonk_amd import Compile, MarshalPublicInputs
from algoplonk_amd._lib import lib, check
from oracle import curves as oc, plonk as oplonk

from helpers import CURVES

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_are_exported_and_bound():
    hdr = open(os.path.join(ROOT, "include", "apk.h")).read()
    declared = set(re.findall(r"\b(apk_[a-z0-9_]+)\s*\(", hdr))
    nm = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    exported = set(re.findall(r" T (apk_[a-z0-9_]+)", nm))
    assert declared, "no declarations parsed"
    assert declared <= exported, "declared but not exported: %s" % sorted(declared - exported)
    assert declared == set(_lib.SYMBOLS), "python binding out of sync: %s" % sorted(declared ^ set(_lib.SYMBOLS))
    assert lib.apk_abi_version() == _lib.ABI_VERSION == 5
    assert lib.apk_g1_bytes(0) == 64 and lib.apk_g1_bytes(1) == 96 and lib.apk_g1_bytes(7) == 0


def test_no_cpu_fallback_when_no_gpu():
    if _lib.device_count() > 0:
        pytest.skip("a GPU is present")
    out = C.create_string_buffer(64)
    cv = ecc.BN254
    rc = lib.apk_g1_mul_batch(cv.abi, 0, cv.g1_to_bytes(cv.g1), cv.fr_vector([3]), 1, out)
    assert rc == _lib.APK_ERR_HIP and b"no HIP device" in lib.apk_last_error()
    d = _lib.CircuitDesc()
    d.curve, d.n = 0, 8
    ctx = C.c_void_p()
    assert lib.apk_ctx_create(C.byref(d), C.byref(ctx)) != 0      # null pointers / no device: an error, never a fallback


@pytest.mark.parametrize("cname", ["bn254", "bls12-381"])
def test_host_field_and_curve_templates_match_oracle(cname):
    cv, ov = CURVES[cname]
    rnd = random.Random(1)
    for field, mod, nb in ((0, cv.r, 32), (1, cv.p, cv.fp_bytes)):
        R = 1 << (8 * nb)
        enc = lambda x: (x * R % mod).to_bytes(nb, "little")
        dec = lambda b: int.from_bytes(b, "little") * pow(R, -1, mod) % mod
        out = C.create_string_buffer(nb)
        cases = [(mod - 1, mod - 1), (mod - 1, 0), (0, 0), (1, mod - 1)] + [(rnd.randrange(mod), rnd.randrange(mod)) for _ in range(100)]
        for a, b in cases:
            for op, exp in ((0, (a + b) % mod), (1, (a - b) % mod), (2, a * b % mod), (4, (-a) % mod)):
                check(lib.apk_host_fe_op(cv.abi, field, op, enc(a), enc(b), out))
                assert dec(out.raw) == exp, (cname, field, op)
            check(lib.apk_host_fe_op(cv.abi, field, 3, enc(a), None, out))
            assert dec(out.raw) == (pow(a, -1, mod) if a else 0)
            # the unsaturated-limb forms (ffu.h: Fp inside the MSM, Fr inside the NTT tiles), converted in and out of gnark's radix
            for op, exp in ((10, a * b % mod), (11, (a + b) % mod), (12, (a - b) % mod), (13, (-a) % mod), (10, a * a % mod)):
                check(lib.apk_host_fe_op(cv.abi, field, op, enc(a), enc(a) if exp == a * a % mod and op == 10 and a != b else enc(b), out))
                assert dec(out.raw) == exp, (cname, field, "unsat", op)
            # 14: ten comparison-free butterfly stages (u, v) <- (u + b v, u - b v) as the NTT tile runs them, u0 = a, v0 = b
            u, v = a, b
            for _ in range(10):
                u, v = (u + b * v) % mod, (u - b * v) % mod
            check(lib.apk_host_fe_op(cv.abi, field, 14, enc(a), enc(b), out))
            assert dec(out.raw) == u, (cname, field, "lazy butterflies")
        # canonical big-endian codecs
        be = C.create_string_buffer(nb)
        check(lib.apk_fe_to_be(cv.abi, field, enc(12345), be))
        assert int.from_bytes(be.raw, "big") == 12345
        check(lib.apk_fe_from_be(cv.abi, field, (mod - 2).to_bytes(nb, "big"), out))
        assert dec(out.raw) == mod - 2
        assert lib.apk_fe_from_be(cv.abi, field, mod.to_bytes(nb, "big"), out) == _lib.APK_ERR_ARG
    for _ in range(4):
        k1, k2 = rnd.randrange(1, cv.r), rnd.randrange(1, cv.r)
        P, Q = ov.mul(ov.g1, k1), ov.mul(ov.g1, k2)
        out = C.create_string_buffer(2 * cv.fp_bytes)
        for op, q, exp in ((0, cv.g1_to_bytes(Q), ov.add(P, Q)), (1, cv.g1_to_bytes(Q), ov.add(P, Q)), (2, None, ov.add(P, P)),
                           (3, cv.fr_to_mont_bytes(k2), ov.mul(P, k2)), (0, cv.g1_to_bytes(P), ov.add(P, P)),
                           (0, cv.g1_to_bytes(ov.neg(P)), None), (1, cv.g1_to_bytes(P), ov.add(P, P)),
                           (1, cv.g1_to_bytes(ov.neg(P)), None), (0, cv.g1_to_bytes(None), P),
                           # 10 / 11: the same additions on unsaturated limbs (dedicated squaring, table-record packing)
                           (10, cv.g1_to_bytes(Q), ov.add(P, Q)), (11, cv.g1_to_bytes(Q), ov.add(P, Q)), (10, cv.g1_to_bytes(P), ov.add(P, P)),
                           (10, cv.g1_to_bytes(ov.neg(P)), None), (11, cv.g1_to_bytes(P), ov.add(P, P)), (11, cv.g1_to_bytes(ov.neg(P)), None),
                           (10, cv.g1_to_bytes(None), P),
                           # 12 / 13: the accumulate loop's lazy mixed addition (no conditional subtractions) vs the plain one over
                           # a 17-step signed chain through doubling, cancellation and an infinity input: a + 3b
                           (12, cv.g1_to_bytes(Q), ov.add(P, ov.mul(Q, 3))), (13, cv.g1_to_bytes(Q), ov.add(P, ov.mul(Q, 3))),
                           (12, cv.g1_to_bytes(P), ov.mul(P, 4)), (12, cv.g1_to_bytes(ov.neg(P)), ov.neg(ov.add(P, P))),
                           # 14: lazy full addition / doubling (equal operands, cancellation, infinity): 4p + 6q
                           (14, cv.g1_to_bytes(Q), ov.add(ov.mul(P, 4), ov.mul(Q, 6))), (14, cv.g1_to_bytes(P), ov.mul(P, 10)),
                           (14, cv.g1_to_bytes(ov.neg(P)), ov.neg(ov.add(P, P)))):
            check(lib.apk_host_g1_op(cv.abi, op, cv.g1_to_bytes(P), q, out))
            assert cv.g1_from_bytes(out.raw) == exp, (cname, op)


@pytest.mark.parametrize("cname", ["bn254", "bls12-381"])
@pytest.mark.parametrize("k", [0, 1, 2])
def test_marshal_proof_layout(cname, k):
    """helper.go:27-88 / MarshalSolidity order and the lengths bsb22_test.go:97-120 asserts."""
    cv, ov = CURVES[cname]
    rnd = random.Random(5)
    pts = [ov.mul(ov.g1, rnd.randrange(1, cv.r)) for _ in range(9 + k)]
    frs = [rnd.randrange(cv.r) for _ in range(7 + k)]
    p = _lib.Proof()
    p.curve, p.nb_commitments = cv.abi, k

    def put(slot, P):
        b = cv.g1_to_bytes(P)
        C.memmove(slot, b, len(b))

    for i in range(3):
        put(p.lro[i], pts[i]); put(p.h[i], pts[3 + i])
    put(p.z, pts[6]); put(p.batched_h, pts[7]); put(p.zshift_h, pts[8])
    for i in range(k):
        put(p.bsb22[i], pts[9 + i])
    for i in range(6 + k):
        C.memmove(p.claimed_values[i], cv.fr_to_mont_bytes(frs[i]), 32)
    C.memmove(p.zshift_value, cv.fr_to_mont_bytes(frs[6 + k]), 32)
    out = C.create_string_buffer(2048)
    n = C.c_size_t(0)
    check(lib.apk_marshal_proof(C.byref(p), out, 2048, C.byref(n)))
    want = oplonk.marshal_proof(ov, oplonk.Proof(lro=pts[0:3], z=pts[6], h=pts[3:6], bsb22_commitments=pts[9:9 + k], batched_h=pts[7],
                                                 claimed_values=frs[: 6 + k], zshift_h=pts[8], zshift_value=frs[6 + k]))
    assert out.raw[: n.value] == want
    base, ptb = (24, 64) if cv is ecc.BN254 else (33, 96)
    assert n.value == base * 32 + k * 32 + k * ptb
    # buffer too small is an error, not a truncation
    assert lib.apk_marshal_proof(C.byref(p), out, 10, C.byref(n)) == _lib.APK_ERR_ARG


def test_hash_fr_matches_template():
    for cname in ("bn254", "bls12-381"):
        cv, ov = CURVES[cname]
        P = ov.mul(ov.g1, 123456789)
        out = C.create_string_buffer(32)
        check(lib.apk_hash_fr(cv.abi, cv.g1_to_bytes(P), out))
        assert cv.fr_from_mont_bytes(out.raw) == oplonk.hash_fr(ov.raw_bytes(P), cv.r)


def test_marshal_public_inputs_and_witness():
    class Sq(frontend.Circuit):
        X = frontend.Public()
        Y = frontend.Secret()

        def define(self, api):
            api.AssertIsEqual(self.X, api.Mul(self.Y, self.Y))

    a = Sq(); a.X, a.Y = 9, 3
    w = frontend.NewWitness(a, ecc.BN254.r)
    assert MarshalPublicInputs(w) == (9).to_bytes(32, "big")
    with pytest.raises(ValueError):
        frontend.NewWitness(Sq(), ecc.BN254.r)


def test_compile_rejects_what_the_reference_rejects():
    """compile_test.go:22-30 (unknown setup), setup/registry_test.go:22-40, algoplonk.go:39-49."""
    class Id(frontend.Circuit):
        X = frontend.Public()

        def define(self, api):
            api.AssertIsEqual(self.X, self.X)

    with pytest.raises(ValueError, match="unknown setup"):
        Compile(Id(), ecc.BN254, 999)
    with pytest.raises(ValueError, match="does not match"):
        Compile(Id(), ecc.BN254, ap_setup.Name.TestOnlyBLS12381)
    with pytest.raises(ValueError, match="unsupported curve"):
        Compile(Id(), ecc.BLS12_377, ap_setup.Name.TestOnlyBN254)
    s, ok = ap_setup.Get(999)
    assert not ok and s is None
    assert ap_setup.Get(ap_setup.Name.DuskBLS12381)[1]
    assert ap_setup.TestOnlySetup(ecc.BN254) == ap_setup.Name.TestOnlyBN254


def test_setup_name_values_are_the_reference_iota_order():
    """setup/setup.go:30-36: an integer setup id carried over from the Go side must name the same setup."""
    N = ap_setup.Name
    assert [int(N.PerpetualPowersOfTauBN254), int(N.EthereumKzgCeremonyBLS12381), int(N.DuskBLS12381), int(N.TestOnlyBN254),
            int(N.TestOnlyBLS12381)] == [0, 1, 2, 3, 4]
    for i, (curve, trusted, path) in enumerate([(ecc.BN254, True, "PerpetualPowersOfTauBN254"),
                                                 (ecc.BLS12_381, True, "EethereumKzgCeremonyBLS12_381"),
                                                 (ecc.BLS12_381, True, "DuskBLS12_381"), (ecc.BN254, False, "test_only"),
                                                 (ecc.BLS12_381, False, "test_only")]):
        s, ok = ap_setup.Get(i)                       # setup/setup.go:49-75
        assert ok and s.Curve is curve and s.Trusted is trusted and s.NamePath == path


def test_frontend_trace_matches_oracle_trace():
    from oracle import circuits as ocircuits

    class Pyth(frontend.Circuit):
        A = frontend.Public(); B = frontend.Public(); C = frontend.Secret()

        def define(self, api):
            api.AssertIsEqual(api.Add(api.Mul(self.A, self.A), api.Mul(self.B, self.B)), api.Mul(self.C, self.C))

    for cname in ("bn254", "bls12-381"):
        cv, ov = CURVES[cname]
        ccs = frontend.Compile(cv.r, Pyth())
        assert ccs.GetNbPublicVariables() == 2 and ccs.GetNbConstraints() == 5 and ccs.domain_size() == 8
        a = Pyth(); a.A, a.B, a.C = 3, 4, 5
        sol = frontend.solve(ccs, frontend.NewWitness(a, cv.r))
        oc_, osol = ocircuits.pythagorean(ov)
        tr, otr = frontend.build_trace(ccs), oplonk.build_trace(oc_)
        # same gates up to the order the builder emitted them: both satisfy the oracle's gate check
        L, R, O = frontend.wire_columns(ccs, sol)
        from helpers import oracle_circuit_from_ccs
        o2 = oracle_circuit_from_ccs(ov, ccs)
        otr2 = oplonk.build_trace(o2)
        assert (tr.ql, tr.qr, tr.qm, tr.qo, tr.qk, tr.perm) == (otr2.ql, otr2.qr, otr2.qm, otr2.qo, otr2.qk, otr2.S)
        assert (L, R, O) == oplonk.solve_lro(o2, sol)
        assert oplonk.check_gates(o2, otr2, L, R, O, sol[:2])


@pytest.mark.parametrize("cname", ["bn254", "bls12-381"])
def test_frontend_commit_path_against_the_oracle(cname):
    """Host logic only: the product's frontend (Commit / AssertIsDifferent, closure-free workload solver, qcp columns,
    commitment indexes) produces a constraint system + witness that the oracle proves and the transcribed verifier accepts."""
    from algoplonk_amd import workloads
    from oracle.prng import tau_from_seed
    cv, ov = CURVES[cname]
    ccs, w, bl, tau = workloads.random_circuit_bsb22(cv, 6, 0xA193, nb_commitments=2, committed=3)
    n = ccs.domain_size()
    assert n == 64 and len(ccs.commitments) == 2
    tr = frontend.build_trace(ccs)
    assert len(tr.qcp) == 2 and all(sum(q) == 3 for q in tr.qcp)
    oc_ = oplonk.Circuit(ov, ccs.GetNbPublicVariables(), ccs.nb_variables, list(ccs.constraints),
                         [oplonk.Commitment(list(rows), cidx) for rows, cidx in ccs.commitments])
    osrs = oplonk.synthetic_srs(ov, n, tau, materialize=False)
    opk = oplonk.setup(oc_, osrs)
    otr = opk.trace
    assert (tr.ql, tr.qr, tr.qm, tr.qo, tr.qk, tr.perm, tr.qcp) == (otr.ql, otr.qr, otr.qm, otr.qo, otr.qk, otr.S, otr.qcp)
    wn = ov.omega(n)
    pi2 = []
    hiding = [(5, 6), (7, 8)]
    sol = frontend.solve(ccs, w, lambda col: oplonk.hash_fr(ov.raw_bytes(osrs.commit(oplonk.intt(col, wn, cv.r))), cv.r), hiding, pi2)
    L, R, O = frontend.wire_columns(ccs, sol)
    assert (L, R, O) == oplonk.solve_lro(oc_, sol)
    pr = oplonk.prove(opk, L, R, O, w.public, bl, pi2=pi2)
    blob, pib = oplonk.marshal_proof(ov, pr), oplonk.marshal_public_inputs(w.public)
    assert oplonk.verify(opk.vk, blob, pib)
    with pytest.raises(ValueError, match="commitment hint"):
        frontend.solve(ccs, w)
