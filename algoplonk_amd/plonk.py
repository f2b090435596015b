"""`plonk.Setup` / `plonk.Prove` as the reference calls them (/root/reference/setup/setup.go:107,149 and
/root/reference/algoplonk.go:89), bound to libapk's C-ABI.  Everything heavy runs in the HIP library; this module
only packs arrays into gnark's in-memory layout and unpacks the proof struct.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

from . import ecc, frontend
from . import _lib
from ._lib import lib, check
from .setup import SRS


@dataclass
class VerifyingKey:
    """Fields the generated verifiers consume (verifier/templateLogicSigBN254.go:21-28,50-72)."""
    curve: ecc.ID
    Size: int
    SizeInv: int
    Generator: int
    CosetShift: int
    NbPublicVariables: int
    Ql: ecc.Point
    Qr: ecc.Point
    Qm: ecc.Point
    Qo: ecc.Point
    Qk: ecc.Point
    S: List[ecc.Point]
    Qcp: List[ecc.Point]
    CommitmentConstraintIndexes: List[int]
    KzgG1: ecc.Point
    tau: Optional[int] = None   # TestOnly setups only: the toxic value (lets tests use the known-tau shortcut)
    KzgG2: Optional[bytes] = None   # Kzg.G2 = ([1]G2, [tau]G2), gnark in-memory G2Affine bytes

    def raw(self) -> "_lib.VerifyingKey":
        """The C-ABI form apk_verify takes (include/apk.h apk_verifying_key)."""
        cv = self.curve
        if not self.KzgG2:
            raise ValueError("verifying key carries no G2 points")
        v = _lib.VerifyingKey()
        v.curve, v.n, v.nb_public, v.nb_commitments = cv.abi, self.Size, self.NbPublicVariables, len(self.Qcp)

        def put(slot, P):
            b = cv.g1_to_bytes(P)
            C.memmove(slot, b, len(b))

        for name, P in (("ql", self.Ql), ("qr", self.Qr), ("qm", self.Qm), ("qo", self.Qo), ("qk", self.Qk), ("g1", self.KzgG1)):
            put(getattr(v, name), P)
        for j in range(3):
            put(v.s[j], self.S[j])
        for k, P in enumerate(self.Qcp):
            put(v.qcp[k], P)
            v.commitment_constraint_index[k] = self.CommitmentConstraintIndexes[k]
        w = 4 * cv.fp_bytes
        for j in range(2):
            C.memmove(v.g2[j], self.KzgG2[j * w: (j + 1) * w], w)
        return v


class ProvingKey:
    """Owns the libapk circuit context (SRS tables, trace polynomials, workspaces resident in HBM)."""

    def __init__(self, curve: ecc.ID, ctx: int, n: int, nb_public: int):
        self.curve = curve
        self._ctx = C.c_void_p(ctx)
        self.n = n
        self.nb_public = nb_public

    @property
    def ctx(self) -> C.c_void_p:
        if not self._ctx:
            raise RuntimeError("proving key was closed")
        return self._ctx

    def close(self) -> None:
        if self._ctx:
            lib.apk_ctx_destroy(self._ctx)
            self._ctx = C.c_void_p(None)

    def __del__(self):  # best effort
        try:
            self.close()
        except Exception:
            pass

    # primitives of SURVEY.md §8a rows a4 / a6, exposed for tests and benchmarks
    def msm(self, scalars: Sequence[int], basis: int = 0) -> ecc.Point:
        cv = self.curve
        buf = cv.fr_vector(scalars)
        out = C.create_string_buffer(2 * cv.fp_bytes)
        check(lib.apk_msm_g1(self.ctx, basis, buf, len(scalars), out))
        return cv.g1_from_bytes(out.raw)

    def ntt(self, values: Sequence[int], which: int = 0, inverse: bool = False, coset: bool = False) -> List[int]:
        cv = self.curve
        buf = C.create_string_buffer(cv.fr_vector(values), len(values) * 32)
        check(lib.apk_ntt(self.ctx, which, int(inverse), int(coset), buf))
        return cv.fr_vector_decode(buf.raw)

    def stats(self, reset: bool = False) -> "_lib.Stats":
        st = _lib.Stats()
        check(lib.apk_stats_read(self.ctx, C.byref(st), int(reset)))
        return st

    def enable_stats(self, on: bool = True) -> None:
        check(lib.apk_stats_enable(self.ctx, int(on)))

    @property
    def msm_window(self) -> int:
        """The signed-digit window width of the context's tables (apk_ctx_msm_window)."""
        return lib.apk_ctx_msm_window(self.ctx)

    def paths(self, reset: bool = False) -> dict:
        """Which forms of the load-dependent kernels the context has taken so far (apk_paths_read)."""
        pc = _lib.PathCounts()
        check(lib.apk_paths_read(self.ctx, C.byref(pc), int(reset)))
        return pc.as_dict()


@dataclass
class Proof:
    """gnark plonk_{bn254,bls12381}.Proof, field for field (helper.go:35-84; bsb22_test.go:71-93)."""
    curve: ecc.ID
    raw: "_lib.Proof"

    def _pt(self, slot) -> ecc.Point:
        return self.curve.g1_from_bytes(bytes(slot))

    def _fr(self, slot) -> int:
        return self.curve.fr_from_mont_bytes(bytes(slot))

    @property
    def LRO(self): return [self._pt(self.raw.lro[i]) for i in range(3)]
    @property
    def Z(self): return self._pt(self.raw.z)
    @property
    def H(self): return [self._pt(self.raw.h[i]) for i in range(3)]
    @property
    def Bsb22Commitments(self): return [self._pt(self.raw.bsb22[i]) for i in range(self.raw.nb_commitments)]
    @property
    def BatchedProofH(self): return self._pt(self.raw.batched_h)
    @property
    def ClaimedValues(self): return [self._fr(self.raw.claimed_values[i]) for i in range(6 + self.raw.nb_commitments)]
    @property
    def ZShiftedOpeningH(self): return self._pt(self.raw.zshift_h)
    @property
    def ZShiftedOpeningClaimedValue(self): return self._fr(self.raw.zshift_value)
    @property
    def challenges(self): return {k: self._fr(getattr(self.raw, k)) for k in ("gamma", "beta", "alpha", "zeta", "gamma_kzg")}


def Setup(ccs: frontend.ConstraintSystem, srs: SRS, device: int = 0, msm_window: int = 0, slots: int = 1):
    """plonk.Setup(ccs, srs, lagrangeSrs) (setup/setup.go:107,149): returns (ProvingKey, VerifyingKey)."""
    cv = srs.curve
    tr = frontend.build_trace(ccs)
    if tr.n != srs.n:
        raise ValueError("SRS sized for n=%d, circuit needs n=%d" % (srs.n, tr.n))
    cols = [cv.fr_vector(c) for c in (tr.ql, tr.qr, tr.qm, tr.qo, tr.qk)]
    perm = (C.c_int64 * len(tr.perm))(*tr.perm)
    d = _lib.CircuitDesc()
    d.curve, d.device, d.n = cv.abi, device, tr.n
    nbc = len(ccs.commitments)
    if nbc > _lib.MAX_COMMITMENTS:
        raise ValueError("at most %d BSB22 commitments" % _lib.MAX_COMMITMENTS)
    if nbc and not srs.g1_lagrange:
        raise ValueError("circuits with BSB22 commitments need the Lagrange SRS")
    d.nb_public, d.nb_commitments = ccs.GetNbPublicVariables(), nbc
    qcp_cols = [cv.fr_vector(c) for c in tr.qcp]
    for k in range(nbc):
        d.qcp[k] = C.cast(C.c_char_p(qcp_cols[k]), C.c_void_p)
        d.commitment_constraint_index[k] = ccs.commitments[k][1]
    keep = [srs.g1, srs.g1_lagrange] + cols + qcp_cols
    d.srs_g1 = C.cast(C.c_char_p(srs.g1), C.c_void_p)
    d.srs_g1_lagrange = C.cast(C.c_char_p(srs.g1_lagrange), C.c_void_p) if srs.g1_lagrange else None
    d.ql, d.qr, d.qm, d.qo, d.qk = (C.cast(C.c_char_p(c), C.c_void_p) for c in cols)
    d.perm = C.cast(perm, C.c_void_p)
    d.msm_window, d.slots = msm_window, slots
    ctx = C.c_void_p()
    check(lib.apk_ctx_create(C.byref(d), C.byref(ctx)))
    del keep
    pk = ProvingKey(cv, ctx.value, tr.n, ccs.GetNbPublicVariables())
    raw = _lib.Vk()
    check(lib.apk_ctx_get_vk(pk.ctx, C.byref(raw)))
    P = lambda slot: cv.g1_from_bytes(bytes(slot))
    F = lambda slot: cv.fr_from_mont_bytes(bytes(slot))
    vk = VerifyingKey(
        curve=cv, Size=tr.n, SizeInv=F(raw.size_inv), Generator=F(raw.generator), CosetShift=F(raw.coset_shift),
        NbPublicVariables=ccs.GetNbPublicVariables(), Ql=P(raw.ql), Qr=P(raw.qr), Qm=P(raw.qm), Qo=P(raw.qo), Qk=P(raw.qk),
        S=[P(raw.s[i]) for i in range(3)], Qcp=[P(raw.qcp[k]) for k in range(nbc)],
        CommitmentConstraintIndexes=[ccs.commitments[k][1] for k in range(nbc)],
        KzgG1=cv.g1_from_bytes(srs.g1[: 2 * cv.fp_bytes]), tau=srs.tau, KzgG2=srs.g2)
    return pk, vk


class VerificationError(RuntimeError):
    pass


def Verify(proof: "Proof", vk: VerifyingKey, public_witness: frontend.Witness) -> None:
    """plonk.Verify(proof, vk, publicWitness) (/root/reference/algoplonk.go:93): raises VerificationError when the proof is
    rejected.  Host-side (apk_verify: transcript, linearised commitment, one two-pair pairing check) - no GPU work."""
    cv = vk.curve
    public = public_witness.Public().public
    if len(public) != vk.NbPublicVariables:       # gnark: "invalid witness size, got %d, expected %d (public)"
        raise VerificationError("invalid witness size, got %d, expected %d (public)" % (len(public), vk.NbPublicVariables))
    pub = cv.fr_vector(public)
    raw_vk = vk.raw()
    rc = lib.apk_verify_ex(C.byref(raw_vk), C.byref(proof.raw), pub, len(public), None)
    if rc == _lib.APK_ERR_VERIFY:
        raise VerificationError((lib.apk_last_error() or b"").decode())
    check(rc)


def solve_with_commitments(ccs: frontend.ConstraintSystem, pk: ProvingKey, witness: frontend.Witness, hiding=None):
    """Round 0 of plonk.Prove for circuits with BSB22 commitments: gnark's solver with its bsb22 hint - kzg.Commit(committed
    column, Lagrange SRS) on the GPU (apk_msm_g1, basis 1), then hash_to_field on the host (apk_hash_fr).  Returns
    (solution, pi2 columns)."""
    cv = pk.curve
    nbc = len(ccs.commitments)
    if hiding is None:
        hiding = [(int.from_bytes(os.urandom(48), "big") % cv.r, int.from_bytes(os.urandom(48), "big") % cv.r) for _ in range(nbc)]

    def commit_hint(col):
        pt = C.create_string_buffer(2 * cv.fp_bytes)
        check(lib.apk_msm_g1(pk.ctx, 1, cv.fr_vector(col), len(col), pt))
        out = C.create_string_buffer(32)
        check(lib.apk_hash_fr(cv.abi, pt, out))
        return cv.fr_from_mont_bytes(out.raw)

    pi2_cols: List[List[int]] = []
    solution = frontend.solve(ccs, witness, commit_hint if nbc else None, hiding, pi2_cols)
    return solution, pi2_cols


def Prove(ccs: frontend.ConstraintSystem, pk: ProvingKey, witness: frontend.Witness,
          blinding: Optional[Sequence[int]] = None, hiding=None) -> Proof:
    """plonk.Prove(ccs, pk, witness) (/root/reference/algoplonk.go:89).  `blinding` = the 9 scalars gnark draws
    from crypto/rand; drawn from os.urandom when omitted."""
    cv = pk.curve
    nbc = len(ccs.commitments)
    solution, pi2_cols = solve_with_commitments(ccs, pk, witness, hiding)
    L, R, O = frontend.wire_columns(ccs, solution)
    if blinding is None:
        blinding = [int.from_bytes(os.urandom(48), "big") % cv.r for _ in range(_lib.NB_BLINDING)]
    if len(blinding) != _lib.NB_BLINDING:
        raise ValueError("need %d blinding scalars" % _lib.NB_BLINDING)
    out = _lib.Proof()
    pi2_bufs = [cv.fr_vector(col) for col in pi2_cols]
    pi2_arr = None
    if nbc:
        pi2_arr = (C.c_void_p * nbc)(*[C.cast(C.c_char_p(b), C.c_void_p) for b in pi2_bufs])
    check(lib.apk_prove(pk.ctx, cv.fr_vector(L), cv.fr_vector(R), cv.fr_vector(O), cv.fr_vector(witness.public),
                        cv.fr_vector(blinding), pi2_arr, C.byref(out)))
    return Proof(cv, out)
