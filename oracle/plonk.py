"""ORACLE (test infrastructure, not product code): PLONK setup / prover / verifier on plain ints.

What it restates
----------------
* The prover AlgoPlonk calls at /root/reference/algoplonk.go:89 (`plonk.Prove`) and the setup it
  calls at setup/setup.go:107,149 (`plonk.Setup`).  Both live in gnark v0.15.0 (go.mod:8), which is
  NOT vendored and not on this machine: the round structure below follows SURVEY.md §3.3 and the
  polynomial identity that the reference's verifier templates force.
* The verifier, transcribed step by step from verifier/templateLogicSigBN254.go:110-356 (BLS twin:
  templateLogicSigBLS12_381.go:124-372), with the final `ec.pairing_check` (:355) replaced by the
  known-tau G1 check for synthetic SRS (SURVEY.md App. E last line).
* The proof / public-input wire formats of helper.go:13-24,27-88,91-110.

PARITY STATUS.  `verify` is PINNED: tests/test_template_pin.py holds its verdicts and every intermediate
(challenges, PI, lin(zeta), [lin], folding challenge, folded digest/claims) to the reference's own
templates, rendered and EXECUTED in the build container (tests/golden/make_template_fixtures.py ->
tests/golden/template_verdicts.json: k = 0,1,2 commitments, both curves, the reference's mutations of
testutils/verifier_integration_test.go:188-228).  `prove` is pinned only through that verifier (it
makes proofs the executed template accepts): its bytes against gnark's for the same randomness are
"parity unpinned" - the reference holds no golden proof bytes and gnark cannot run here.
Also pinned: blob shape/offsets (bsb22_test.go:70,83,97-120).

The polynomial arithmetic here is deliberately the *textbook* route (numerator on an 8n domain, exact
division by X^n-1 with a zero-remainder check) so it is independent from the 4n-coset schedule the HIP
path uses.
"""
from __future__ import annotations

import hashlib
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

from .curves import Curve, Point

# ------------------------------------------------------------------------------------------------
# polynomials over Fr
# ------------------------------------------------------------------------------------------------


def bitrev_permute(a: list) -> list:
    n = len(a)
    k = n.bit_length() - 1
    out = [0] * n
    for i in range(n):
        out[int(format(i, "0%db" % k)[::-1], 2) if k else 0] = a[i]
    return out


def ntt(a: Sequence[int], omega: int, r: int) -> List[int]:
    """Natural-order in, natural-order out: A[k] = sum_j a[j] omega^(jk)."""
    n = len(a)
    a = bitrev_permute(list(a))
    length = 2
    while length <= n:
        w_len = pow(omega, n // length, r)
        half = length // 2
        for start in range(0, n, length):
            w = 1
            for j in range(half):
                u = a[start + j]
                v = a[start + j + half] * w % r
                a[start + j] = (u + v) % r
                a[start + j + half] = (u - v) % r
                w = w * w_len % r
        length *= 2
    return a


def intt(a: Sequence[int], omega: int, r: int) -> List[int]:
    n = len(a)
    ninv = pow(n, -1, r)
    return [x * ninv % r for x in ntt(a, pow(omega, -1, r), r)]


def poly_eval(c: Sequence[int], x: int, r: int) -> int:
    acc = 0
    for v in reversed(c):
        acc = (acc * x + v) % r
    return acc


def poly_div_linear(c: Sequence[int], z: int, r: int) -> List[int]:
    """(c(X) - c(z)) / (X - z) by synthetic division (kzg.Open [UPSTREAM gnark-crypto])."""
    out = [0] * (len(c) - 1)
    acc = 0
    for i in range(len(c) - 1, 0, -1):
        acc = (acc * z + c[i]) % r
        out[i - 1] = acc
    return out


# ------------------------------------------------------------------------------------------------
# circuit (the host-side object gnark's frontend hands over: cs.SparseR1CS, algoplonk.go:50)
# ------------------------------------------------------------------------------------------------


@dataclass
class Commitment:
    """BSB22 commitment info (gnark constraint.PlonkCommitment [UPSTREAM]).  `committed` = constraint
    indexes whose L wire is committed; `commitment_index` = constraint index that receives the hash
    (VK `CommitmentConstraintIndexes`, templateLogicSigBN254.go:187-193)."""
    committed: List[int]
    commitment_index: int


@dataclass
class Circuit:
    """gate i:  ql*a + qr*b + qm*a*b + qo*c + qk == 0  with a,b,c = value of wires xa,xb,xc."""
    curve: Curve
    nb_public: int
    nb_variables: int
    constraints: List[Tuple[int, int, int, int, int, int, int, int]] = field(default_factory=list)
    commitments: List[Commitment] = field(default_factory=list)

    def domain_size(self) -> int:
        m = self.nb_public + len(self.constraints)
        n = 1
        while n < m:
            n *= 2
        return max(n, 8)  # keep n >= 8 so deg h < 4n (gnark switches to an 8n domain below 6)


@dataclass
class Trace:
    n: int
    ql: List[int]
    qr: List[int]
    qm: List[int]
    qo: List[int]
    qk: List[int]
    qcp: List[List[int]]
    S: List[int]  # permutation over 3n cells (gnark trace.S)


def build_trace(c: Circuit) -> Trace:
    """gnark plonk.NewTrace + buildPermutation [UPSTREAM, restated from memory, SURVEY.md §3.1].
    Public rows are `ql = -1` placeholders (SURVEY.md App. E note)."""
    r = c.curve.r
    n = c.domain_size()
    ql = [0] * n
    qr = [0] * n
    qm = [0] * n
    qo = [0] * n
    qk = [0] * n
    for i in range(c.nb_public):
        ql[i] = r - 1
    off = c.nb_public
    lro = [0] * (3 * n)
    for i in range(c.nb_public):
        lro[i] = i
    for j, (a, b, m, o, k, xa, xb, xc) in enumerate(c.constraints):
        ql[off + j], qr[off + j], qm[off + j], qo[off + j], qk[off + j] = a % r, b % r, m % r, o % r, k % r
        lro[off + j] = xa
        lro[n + off + j] = xb
        lro[2 * n + off + j] = xc
    qcp = []
    for cm in c.commitments:
        q = [0] * n
        for ci in cm.committed:
            q[off + ci] = 1
        qcp.append(q)
    perm = [-1] * (3 * n)
    cycle = [-1] * c.nb_variables
    for i in range(3 * n):
        if cycle[lro[i]] != -1:
            perm[i] = cycle[lro[i]]
        cycle[lro[i]] = i
    for i in range(3 * n):
        if perm[i] == -1:
            perm[i] = cycle[lro[i]]
    return Trace(n, ql, qr, qm, qo, qk, qcp, perm)


def solve_lro(c: Circuit, solution: Sequence[int]) -> Tuple[List[int], List[int], List[int]]:
    """Wire columns from a full variable assignment (gnark evaluateLROSmallDomain [UPSTREAM]):
    placeholder and padding cells carry variable 0 so they sit in variable 0's permutation cycle."""
    n = c.domain_size()
    s0 = solution[0]
    L = [s0] * n
    R = [s0] * n
    O = [s0] * n
    for i in range(c.nb_public):
        L[i] = solution[i]
    off = c.nb_public
    for j, (_, _, _, _, _, xa, xb, xc) in enumerate(c.constraints):
        L[off + j], R[off + j], O[off + j] = solution[xa], solution[xb], solution[xc]
    return L, R, O


def check_gates(c: Circuit, tr: Trace, L, R, O, public: Sequence[int], qk_completed=None) -> bool:
    r = c.curve.r
    qk = list(tr.qk) if qk_completed is None else qk_completed
    if qk_completed is None:
        for i in range(c.nb_public):
            qk[i] = public[i]
    for i in range(tr.n):
        v = tr.ql[i] * L[i] + tr.qr[i] * R[i] + tr.qm[i] * L[i] * R[i] + tr.qo[i] * O[i] + qk[i]
        if v % r:
            return False
    return True


# ------------------------------------------------------------------------------------------------
# SRS + setup
# ------------------------------------------------------------------------------------------------


@dataclass
class SRS:
    """KZG SRS in G1: canonical [tau^i]G1 (n+3 points, setup/setup.go:113-114) and Lagrange
    [L_i(tau)]G1 (n points, setup/setup.go:124,138).  `tau` kept only for synthetic SRS."""
    curve: Curve
    g1: List[Point]
    g1_lagrange: List[Point]
    tau: Optional[int] = None

    def commit(self, coeffs: Sequence[int]) -> Point:
        """kzg.Commit over the canonical SRS.  With a known tau this is f(tau)*G1 (same group
        element as the MSM); otherwise the naive MSM."""
        cv = self.curve
        if self.tau is not None:
            return cv.mul(cv.g1, poly_eval(coeffs, self.tau, cv.r))
        assert len(coeffs) <= len(self.g1)
        return cv.msm_naive(self.g1[: len(coeffs)], coeffs)

    def commit_lagrange(self, evals: Sequence[int]) -> Point:
        cv = self.curve
        assert len(evals) == len(self.g1_lagrange)
        return cv.msm_naive(self.g1_lagrange, evals)


def synthetic_srs(curve: Curve, n: int, tau: int, materialize: bool = True) -> SRS:
    """gnark test/unsafekzg analogue (setup/setup.go:103): SRS from a known tau."""
    r = curve.r
    g1: List[Point] = []
    lag: List[Point] = []
    if materialize:
        t = 1
        for _ in range(n + 3):
            g1.append(curve.mul(curve.g1, t))
            t = t * tau % r
        # L_i(tau) = omega^i (tau^n - 1) / (n (tau - omega^i))
        w = curve.omega(n)
        zn = (pow(tau, n, r) - 1) * pow(n, -1, r) % r
        wi = 1
        for _ in range(n):
            lag.append(curve.mul(curve.g1, wi * zn % r * pow(tau - wi, -1, r) % r))
            wi = wi * w % r
    return SRS(curve, g1, lag, tau)


@dataclass
class VerifyingKey:
    """Fields the templates consume (templateLogicSigBN254.go:21-28,50-72)."""
    curve: Curve
    size: int
    size_inv: int
    generator: int
    coset_shift: int
    nb_public: int
    ql: Point
    qr: Point
    qm: Point
    qo: Point
    qk: Point
    s: List[Point]
    qcp: List[Point]
    commitment_constraint_indexes: List[int]
    g1: Point
    tau: Optional[int] = None  # synthetic SRS only: replaces G2 = ([1]G2, [tau]G2)
    g2: Optional[tuple] = None  # (G2_SRS_0, G2_SRS_1): finish with the real pairing check (pairing_bls12381.py / pairing_bn254.py)


@dataclass
class ProvingKey:
    curve: Curve
    trace: Trace
    srs: SRS
    vk: VerifyingKey
    # canonical forms of the trace polynomials
    ql_c: List[int]
    qr_c: List[int]
    qm_c: List[int]
    qo_c: List[int]
    qk_c: List[int]
    s_c: List[List[int]]
    qcp_c: List[List[int]]
    s_lag: List[List[int]]


def setup(c: Circuit, srs: SRS) -> ProvingKey:
    """plonk.Setup (setup/setup.go:107,149) [UPSTREAM]: trace -> canonical polys -> 8+k commitments."""
    cv = c.curve
    r = cv.r
    tr = build_trace(c)
    n = tr.n
    w = cv.omega(n)
    u = cv.coset_shift
    ids = []
    wi = 1
    for _ in range(n):
        ids.append(wi)
        wi = wi * w % r
    ids = ids + [x * u % r for x in ids] + [x * u % r * u % r for x in ids]
    s_lag = [[ids[tr.S[j * n + i]] for i in range(n)] for j in range(3)]
    can = lambda v: intt(v, w, r)
    ql_c, qr_c, qm_c, qo_c, qk_c = can(tr.ql), can(tr.qr), can(tr.qm), can(tr.qo), can(tr.qk)
    s_c = [can(s) for s in s_lag]
    qcp_c = [can(q) for q in tr.qcp]
    vk = VerifyingKey(
        curve=cv, size=n, size_inv=pow(n, -1, r), generator=w, coset_shift=u, nb_public=c.nb_public,
        ql=srs.commit(ql_c), qr=srs.commit(qr_c), qm=srs.commit(qm_c), qo=srs.commit(qo_c),
        qk=srs.commit(qk_c), s=[srs.commit(s) for s in s_c], qcp=[srs.commit(q) for q in qcp_c],
        commitment_constraint_indexes=[cm.commitment_index for cm in c.commitments],
        g1=srs.g1[0] if srs.g1 else cv.g1, tau=srs.tau,
    )
    return ProvingKey(cv, tr, srs, vk, ql_c, qr_c, qm_c, qo_c, qk_c, s_c, qcp_c, s_lag)


# ------------------------------------------------------------------------------------------------
# Fiat-Shamir (gnark-crypto fiat-shamir + fr.Hash [UPSTREAM]; spec mirrored by the templates)
# ------------------------------------------------------------------------------------------------


def fs_challenge(name: bytes, prev: Optional[bytes], *bindings: bytes) -> bytes:
    """templateLogicSigBN254.go:131-135: sha256(name || previous raw digest || bindings)."""
    h = hashlib.sha256()
    h.update(name)
    if prev is not None:
        h.update(prev)
    for b in bindings:
        h.update(b)
    return h.digest()


def hash_fr(point_bytes: bytes, r: int) -> int:
    """templateLogicSigBN254.go:386-397 (BLS twin :410-420): expand_msg_xmd(sha256) with DST
    'BSB22-Plonk', 48 output bytes reduced mod r."""
    dst_prime = b"BSB22-Plonk\x0b"
    b0 = hashlib.sha256(bytes(64) + point_bytes + b"\x00\x30\x00" + dst_prime).digest()
    b1 = hashlib.sha256(b0 + b"\x01" + dst_prime).digest()
    b2 = hashlib.sha256(bytes(x ^ y for x, y in zip(b0, b1)) + b"\x02" + dst_prime).digest()
    return (int.from_bytes(b1, "big") * (1 << 128) + int.from_bytes(b2[:16], "big")) % r


def fr_bytes(x: int) -> bytes:
    return x.to_bytes(32, "big")


# ------------------------------------------------------------------------------------------------
# proof container + wire format
# ------------------------------------------------------------------------------------------------


@dataclass
class Proof:
    """gnark plonk_{bn254,bls12381}.Proof field-for-field (helper.go:35-84; bsb22_test.go:71-93)."""
    lro: List[Point]
    z: Point
    h: List[Point]
    bsb22_commitments: List[Point]
    batched_h: Point
    claimed_values: List[int]          # lin, l, r, o, s1, s2, qcp_i...
    zshift_h: Point
    zshift_value: int


def marshal_proof(cv: Curve, pr: Proof) -> bytes:
    """helper.go:13-24 (BN254 = gnark MarshalSolidity, same order: templateLogicSigBN254.go:75-108)
    and helper.go:27-88 (BLS12-381)."""
    rb = cv.raw_bytes
    out = b"".join(rb(p) for p in pr.lro)
    out += b"".join(rb(p) for p in pr.h)
    out += b"".join(fr_bytes(v) for v in pr.claimed_values[1:6])
    out += rb(pr.z)
    out += fr_bytes(pr.zshift_value)
    out += rb(pr.batched_h)
    out += rb(pr.zshift_h)
    k = len(pr.bsb22_commitments)
    out += b"".join(fr_bytes(pr.claimed_values[6 + i]) for i in range(k))
    out += b"".join(rb(p) for p in pr.bsb22_commitments)
    return out


def marshal_public_inputs(public: Sequence[int]) -> bytes:
    """helper.go:91-110: witness MarshalBinary minus the 12-byte header."""
    return b"".join(fr_bytes(v) for v in public)


# ------------------------------------------------------------------------------------------------
# prover
# ------------------------------------------------------------------------------------------------


def _blind(can: List[int], b: Sequence[int], n: int, r: int) -> List[int]:
    """p(X) + b(X)(X^n - 1)   (gnark commitToPolyAndBlinding / commitBlindingFactor [UPSTREAM])."""
    out = list(can) + [0] * len(b)
    for i, bi in enumerate(b):
        out[i] = (out[i] - bi) % r
        out[n + i] = (out[n + i] + bi) % r
    return out


def _padmul_ntt(polys: Sequence[Sequence[int]], N: int, w: int, r: int) -> List[List[int]]:
    return [ntt(list(p) + [0] * (N - len(p)), w, r) for p in polys]


@dataclass
class ProverTrace:
    """Intermediate values exposed for stage-by-stage parity checks against the HIP path."""
    gamma: int = 0
    beta: int = 0
    alpha: int = 0
    zeta: int = 0
    gamma_kzg: int = 0
    z_lagrange: List[int] = field(default_factory=list)
    h_coeffs: List[int] = field(default_factory=list)
    lin_coeffs: List[int] = field(default_factory=list)
    blinded: dict = field(default_factory=dict)


def prove(pk: ProvingKey, L: Sequence[int], R: Sequence[int], O: Sequence[int],
          public: Sequence[int], blinding: Sequence[int],
          pi2: Sequence[Sequence[int]] = (), trace_out: Optional[ProverTrace] = None) -> Proof:
    """plonk.Prove (algoplonk.go:89) [UPSTREAM gnark v0.15.0 backend/plonk/<curve>/prove.go].

    Inputs are what gnark's host side holds after `solveConstraints`: the wire columns L,R,O in
    Lagrange form (size n), the public inputs, and - explicit here, crypto/rand upstream - the 9
    blinding scalars (bl0,bl1, br0,br1, bo0,bo1, bz0,bz1,bz2).  `pi2` are the
    BSB22 committed columns (Lagrange, hiding entries already placed).
    """
    cv = pk.curve
    r = cv.r
    tr = pk.trace
    n = tr.n
    w = cv.omega(n)
    u = cv.coset_shift
    vk = pk.vk
    srs = pk.srs
    rb = cv.raw_bytes
    T = trace_out if trace_out is not None else ProverTrace()
    assert len(L) == len(R) == len(O) == n and len(blinding) == 9 and len(public) == vk.nb_public
    nb_c = len(tr.qcp)
    assert len(pi2) == nb_c

    # ---- round 1: BSB22 commitments, blinded wire commitments, completed Qk ------------------
    bsb = [srs.commit_lagrange(p) if srs.tau is None else srs.commit(intt(p, w, r)) for p in pi2]
    cvals = [hash_fr(rb(P), r) for P in bsb]
    l_c, r_c, o_c = intt(L, w, r), intt(R, w, r), intt(O, w, r)
    bl = _blind(l_c, blinding[0:2], n, r)
    br = _blind(r_c, blinding[2:4], n, r)
    bo = _blind(o_c, blinding[4:6], n, r)
    lro = [srs.commit(bl), srs.commit(br), srs.commit(bo)]
    qk_lag = list(tr.qk)
    for i in range(vk.nb_public):
        qk_lag[i] = public[i] % r
    for i, ci in enumerate(vk.commitment_constraint_indexes):
        qk_lag[vk.nb_public + ci] = cvals[i]
    qk_full = intt(qk_lag, w, r)
    pi2_c = [intt(p, w, r) for p in pi2]

    # ---- gamma, beta (templateLogicSigBN254.go:131-133) ---------------------------------------
    gamma_raw = fs_challenge(
        b"gamma", None,
        rb(vk.s[0]), rb(vk.s[1]), rb(vk.s[2]), rb(vk.ql), rb(vk.qr), rb(vk.qm), rb(vk.qo), rb(vk.qk),
        *[rb(q) for q in vk.qcp], marshal_public_inputs(public), rb(lro[0]), rb(lro[1]), rb(lro[2]))
    beta_raw = fs_challenge(b"beta", gamma_raw)
    gamma = int.from_bytes(gamma_raw, "big") % r
    beta = int.from_bytes(beta_raw, "big") % r

    # ---- round 2: grand product (SURVEY.md App. E "grand product rows") -----------------------
    Z = [1] * n
    wi = 1
    wires = (L, R, O)
    for i in range(n - 1):
        num = den = 1
        for j in range(3):
            num = num * (wires[j][i] + beta * pow(u, j, r) * wi + gamma) % r
            den = den * (wires[j][i] + beta * pk.s_lag[j][i] + gamma) % r
        Z[i + 1] = Z[i] * num % r * pow(den, -1, r) % r
        wi = wi * w % r
    z_c = intt(Z, w, r)
    bz = _blind(z_c, blinding[6:9], n, r)
    z_com = srs.commit(bz)

    alpha_raw = fs_challenge(b"alpha", beta_raw, *[rb(P) for P in bsb], rb(z_com))
    alpha = int.from_bytes(alpha_raw, "big") % r

    # ---- round 3: quotient ----------------------------------------------------------------------
    N = 8 * n
    W = cv.omega(N)
    zs = [c * pow(w, i, r) % r for i, c in enumerate(bz)]            # Z(omega X)
    x1 = [0, 1]
    # L_0(X) = (X^n - 1) / (n (X - 1)) = (1/n) sum_i X^i
    ninv = pow(n, -1, r)
    L0 = [ninv] * n
    polys = [bl, br, bo, bz, zs, pk.ql_c, pk.qr_c, pk.qm_c, pk.qo_c, qk_full,
             pk.s_c[0], pk.s_c[1], pk.s_c[2], x1, L0] + list(pk.qcp_c) + list(pi2_c)
    ev = _padmul_ntt(polys, N, W, r)
    (e_l, e_r, e_o, e_z, e_zs, e_ql, e_qr, e_qm, e_qo, e_qk, e_s1, e_s2, e_s3, e_x, e_l0) = ev[:15]
    e_qcp = ev[15:15 + nb_c]
    e_pi2 = ev[15 + nb_c:]
    a2 = alpha * alpha % r
    num_ev = [0] * N
    for i in range(N):
        l_, r_, o_ = e_l[i], e_r[i], e_o[i]
        gate = (e_ql[i] * l_ + e_qr[i] * r_ + e_qm[i] * l_ * r_ + e_qo[i] * o_ + e_qk[i]) % r
        for k in range(nb_c):
            gate = (gate + e_qcp[k][i] * e_pi2[k][i]) % r
        x = e_x[i]
        pa = e_zs[i] * (l_ + beta * e_s1[i] + gamma) % r * (r_ + beta * e_s2[i] + gamma) % r \
            * (o_ + beta * e_s3[i] + gamma) % r
        pb = e_z[i] * (l_ + beta * x + gamma) % r * (r_ + beta * u * x + gamma) % r \
            * (o_ + beta * u * u * x + gamma) % r
        loc = e_l0[i] * (e_z[i] - 1) % r
        num_ev[i] = (gate + alpha * (pa - pb) + a2 * loc) % r
    num = intt(num_ev, W, r)
    # exact division by X^n - 1
    top = len(num)
    while top > 0 and num[top - 1] == 0:
        top -= 1
    hq = [0] * max(top - n, 0)
    for j in range(top - n - 1, -1, -1):
        hq[j] = (num[j + n] + (hq[j + n] if j + n < len(hq) else 0)) % r
    for j in range(n):
        assert (num[j] + (hq[j] if j < len(hq) else 0)) % r == 0, "quotient not exact: witness does not satisfy the circuit"
    assert len(hq) <= 3 * (n + 2)
    hq = hq + [0] * (3 * (n + 2) - len(hq))
    h1, h2, h3 = hq[: n + 2], hq[n + 2: 2 * (n + 2)], hq[2 * (n + 2):]
    hcom = [srs.commit(h1), srs.commit(h2), srs.commit(h3)]

    zeta_raw = fs_challenge(b"zeta", alpha_raw, rb(hcom[0]), rb(hcom[1]), rb(hcom[2]))
    zeta = int.from_bytes(zeta_raw, "big") % r

    # ---- round 4: openings ----------------------------------------------------------------------
    zw = zeta * w % r
    z_shift_val = poly_eval(bz, zw, r)
    zshift_h = srs.commit(poly_div_linear(bz, zw, r))

    lz, rz, oz = poly_eval(bl, zeta, r), poly_eval(br, zeta, r), poly_eval(bo, zeta, r)
    s1z, s2z = poly_eval(pk.s_c[0], zeta, r), poly_eval(pk.s_c[1], zeta, r)
    qcpz = [poly_eval(q, zeta, r) for q in pk.qcp_c]
    zn = (pow(zeta, n, r) - 1) % r
    lag0 = zn * pow(n, -1, r) % r * pow(zeta - 1, -1, r) % r
    # coefficients of S3(X) and Z(X) in lin(X)  (templateLogicSigBN254.go:232-254)
    c_s3 = alpha * beta % r * z_shift_val % r * (lz + beta * s1z + gamma) % r * (rz + beta * s2z + gamma) % r
    c_z = (-alpha * (lz + beta * zeta + gamma) % r * (rz + beta * u * zeta + gamma) % r
           * (oz + beta * u * u * zeta + gamma) + a2 * lag0) % r
    zn2 = pow(zeta, n + 2, r)
    lin = [0] * (n + 3)
    for i in range(n + 3):
        t = 0
        if i < n:
            t = (lz * pk.ql_c[i] + rz * pk.qr_c[i] + lz * rz % r * pk.qm_c[i] + oz * pk.qo_c[i]
                 + pk.qk_c[i] + c_s3 * pk.s_c[2][i]) % r
            for k in range(nb_c):
                t = (t + qcpz[k] * pi2_c[k][i]) % r
        t = (t + c_z * bz[i]) % r
        if i < n + 2:
            t = (t - zn * (h1[i] + zn2 * h2[i] + zn2 * zn2 % r * h3[i])) % r
        lin[i] = t
    lin_val = poly_eval(lin, zeta, r)
    lin_com = srs.commit(lin)

    # kzg.BatchOpenSinglePoint [UPSTREAM]; transcript = templateLogicSigBN254.go:280-286
    polys_open = [lin, bl, br, bo, pk.s_c[0], pk.s_c[1]] + list(pk.qcp_c)
    digests = [lin_com, lro[0], lro[1], lro[2], vk.s[0], vk.s[1]] + list(vk.qcp)
    claimed = [lin_val, lz, rz, oz, s1z, s2z] + qcpz
    g_raw = fs_challenge(b"gamma", None, fr_bytes(zeta), *[rb(d) for d in digests],
                         *[fr_bytes(v) for v in claimed], fr_bytes(z_shift_val))
    gk = int.from_bytes(g_raw, "big") % r
    folded = [0] * (n + 3)
    acc = 1
    for p in polys_open:
        for i, c in enumerate(p):
            folded[i] = (folded[i] + acc * c) % r
        acc = acc * gk % r
    batched_h = srs.commit(poly_div_linear(folded, zeta, r))

    T.gamma, T.beta, T.alpha, T.zeta, T.gamma_kzg = gamma, beta, alpha, zeta, gk
    T.z_lagrange, T.h_coeffs, T.lin_coeffs = Z, hq, lin
    T.blinded = {"l": bl, "r": br, "o": bo, "z": bz}
    return Proof(lro=lro, z=z_com, h=hcom, bsb22_commitments=bsb, batched_h=batched_h,
                 claimed_values=claimed, zshift_h=zshift_h, zshift_value=z_shift_val)


# ------------------------------------------------------------------------------------------------
# verifier: transcription of verifier/templateLogicSigBN254.go (line numbers in comments)
# ------------------------------------------------------------------------------------------------


def verify(vk: VerifyingKey, proof: bytes, public_inputs: bytes, trace_out: Optional[dict] = None) -> bool:
    """`trace_out`, when given, receives the intermediates the template computes under the template's own names
    (gamma, beta, alpha, zeta, PI, linearized_poly_at_z, lin_poly_com, folded_h, digest, claims, quotient) so that
    tests/test_template_pin.py can hold every step - not only the verdict - to the executed reference template."""
    T = trace_out if trace_out is not None else {}
    cv = vk.curve
    q = cv.r
    fpb = cv.fp_bytes
    pt = 2 * fpb
    k = len(vk.commitment_constraint_indexes)
    base = 24 * 32 if fpb == 32 else 33 * 32
    # :50-51 length checks
    if len(proof) != base + k * (32 + pt) or len(public_inputs) != vk.nb_public * 32:
        return False
    pos = 0

    def take(nbytes):
        nonlocal pos
        b = proof[pos: pos + nbytes]
        pos += nbytes
        return b

    # :75-108 read proof
    L_COM, R_COM, O_COM = take(pt), take(pt), take(pt)
    H_0, H_1, H_2 = take(pt), take(pt), take(pt)
    L_AT_Z, R_AT_Z, O_AT_Z, S1_AT_Z, S2_AT_Z = (take(32) for _ in range(5))
    GRAND_PRODUCT = take(pt)
    GRAND_PRODUCT_AT_Z_OMEGA = take(32)
    BATCH_OPENING_AT_Z = take(pt)
    OPENING_AT_Z_OMEGA = take(pt)
    QCP_AT_Z = [take(32) for _ in range(k)]
    BSB_COM = [take(pt) for _ in range(k)]
    I = lambda b: int.from_bytes(b, "big")
    # :110-124 well-formedness
    for b in [L_AT_Z, R_AT_Z, O_AT_Z, S1_AT_Z, S2_AT_Z, GRAND_PRODUCT_AT_Z_OMEGA] + QCP_AT_Z:
        if I(b) >= q:
            return False
    pub = [I(public_inputs[i * 32:(i + 1) * 32]) for i in range(vk.nb_public)]
    if any(v >= q for v in pub):
        return False

    rb = cv.raw_bytes

    def fs(p: bytes) -> bytes:
        # templateLogicSigBLS12_381.go:402-407 (BN254 template hashes the bytes as they are)
        if fpb == 48 and p == bytes(96):
            return bytes([0x80]) + p[1:]
        return p

    def P(b: bytes) -> Point:
        pnt = cv.from_raw_bytes(b)
        if not cv.is_on_curve(pnt):
            raise ValueError("point not on curve")  # the AVM ec ops fail the program
        return pnt

    try:
        # :131-140 challenges
        gamma_pre = hashlib.sha256(
            b"gamma" + rb(vk.s[0]) + rb(vk.s[1]) + rb(vk.s[2]) + rb(vk.ql) + rb(vk.qr) + rb(vk.qm)
            + rb(vk.qo) + rb(vk.qk) + b"".join(rb(x) for x in vk.qcp) + public_inputs
            + fs(L_COM) + fs(R_COM) + fs(O_COM)).digest()
        beta_pre = hashlib.sha256(b"beta" + gamma_pre).digest()
        alpha_pre = hashlib.sha256(b"alpha" + beta_pre + b"".join(fs(b) for b in BSB_COM)
                                   + fs(GRAND_PRODUCT)).digest()
        zeta_pre = hashlib.sha256(b"zeta" + alpha_pre + fs(H_0) + fs(H_1) + fs(H_2)).digest()
        gamma, beta, alpha, zeta = (I(x) % q for x in (gamma_pre, beta_pre, alpha_pre, zeta_pre))
        T.update(gamma=gamma, beta=beta, alpha=alpha, zeta=zeta)

        n = vk.size
        # :142-146
        Zz = (pow(zeta, n, q) + q - 1) % q
        zn = Zz * vk.size_inv % q
        # :148-186 public-input interpolation (batch inversion collapsed to per-element inverses)
        PI = 0
        w_ = 1
        for i in range(vk.nb_public):
            x = (zeta + q - w_) % q
            if x == 0:
                return False  # expmod(0, q-2) = 0 upstream -> proof would fail anyway
            li = w_ * (pow(x, -1, q) * zn % q) % q
            PI = (PI + li * pub[i]) % q
            w_ = w_ * vk.generator % q
        # :187-193 BSB22 contributions
        for i, ci in enumerate(vk.commitment_constraint_indexes):
            w_pow = pow(vk.generator, vk.nb_public + ci, q)
            tmp = pow((zeta + q - w_pow) % q, q - 2, q)
            tmp = tmp * (w_pow * zn % q) % q
            PI = (PI + hash_fr(fs(BSB_COM[i]), q) * tmp) % q
        # :195-201
        res = pow((zeta + q - 1) % q, q - 2, q)
        alpha2Lagrange = res * zn % q * alpha % q * alpha % q
        # :203-218
        lz, rz, oz, s1z, s2z, zwz = (I(b) for b in (L_AT_Z, R_AT_Z, O_AT_Z, S1_AT_Z, S2_AT_Z,
                                                     GRAND_PRODUCT_AT_Z_OMEGA))
        s1 = (s1z * beta + gamma + lz) % q
        s2 = (s2z * beta + gamma + rz) % q
        o = (oz + gamma) % q
        s1 = s1 * s2 % q * o % q * alpha % q * zwz % q
        s1 = (s1 + PI + q - alpha2Lagrange) % q
        linearized_poly_at_z = (q - s1) % q
        T.update(PI=PI, linearized_poly_at_z=linearized_poly_at_z)
        # :220-229 folded H
        zn2 = pow(zeta, n + 2, q)
        folded_h = cv.mul(P(H_2), zn2)
        folded_h = cv.add(folded_h, P(H_1))
        folded_h = cv.mul(folded_h, zn2)
        folded_h = cv.add(folded_h, P(H_0))
        folded_h = cv.neg(cv.mul(folded_h, Zz))
        T["folded_h"] = rb_ec(cv, folded_h)
        # :231-254
        uu = zwz * beta % q
        v = (s1z * beta + lz + gamma) % q
        w = (s2z * beta + rz + gamma) % q
        s1 = uu * v % q * w % q * alpha % q
        coset_square = vk.coset_shift * vk.coset_shift % q
        betazeta = beta * zeta % q
        uu = (betazeta + lz + gamma) % q
        v = (betazeta * vk.coset_shift + rz + gamma) % q
        w = (betazeta * coset_square + oz + gamma) % q
        s2 = (q - uu * v % q * w % q) % q
        s2 = (s2 * alpha + alpha2Lagrange) % q
        # :256-278 linearised commitment
        lin = cv.mul(vk.ql, lz)
        lin = cv.add(lin, cv.mul(vk.qr, rz))
        lin = cv.add(lin, cv.mul(vk.qo, oz))
        lin = cv.add(lin, cv.mul(vk.qm, lz * rz % q))
        lin = cv.add(lin, vk.qk)
        for i in range(k):
            lin = cv.add(lin, cv.mul(P(BSB_COM[i]), I(QCP_AT_Z[i])))
        lin = cv.add(lin, cv.mul(vk.s[2], s1))
        lin = cv.add(lin, cv.mul(P(GRAND_PRODUCT), s2))
        lin = cv.add(lin, folded_h)
        T["lin_poly_com"] = rb_ec(cv, lin)
        # :280-287
        r_pre = hashlib.sha256(
            b"gamma" + fr_bytes(zeta) + rb_ec(cv, lin) + fs(L_COM) + fs(R_COM) + fs(O_COM)
            + rb(vk.s[0]) + rb(vk.s[1]) + b"".join(rb(x) for x in vk.qcp)
            + fr_bytes(linearized_poly_at_z) + L_AT_Z + R_AT_Z + O_AT_Z + S1_AT_Z + S2_AT_Z
            + b"".join(QCP_AT_Z) + GRAND_PRODUCT_AT_Z_OMEGA).digest()
        rr = I(r_pre) % q
        r_acc = rr
        T["gamma_kzg"] = rr
        # :289-320 fold
        digest = lin
        claims = linearized_poly_at_z
        for com, val in ((P(L_COM), lz), (P(R_COM), rz), (P(O_COM), oz), (vk.s[0], s1z), (vk.s[1], s2z)):
            digest = cv.add(digest, cv.mul(com, r_acc))
            claims = (claims + val * r_acc) % q
            r_acc = r_acc * rr % q
        for i in range(k):
            digest = cv.add(digest, cv.mul(vk.qcp[i], r_acc))
            claims = (claims + I(QCP_AT_Z[i]) * r_acc) % q
            r_acc = r_acc * rr % q
        T.update(folded_digest=rb_ec(cv, digest), folded_claims=claims)
        # :322-345 batch the two openings with verifier-side randomness
        r_pre = hashlib.sha256(rb_ec(cv, digest) + BATCH_OPENING_AT_Z + fs(GRAND_PRODUCT)
                               + OPENING_AT_Z_OMEGA + fr_bytes(zeta) + fr_bytes(rr)).digest()
        rv = I(r_pre) % q
        quotient = cv.add(P(BATCH_OPENING_AT_Z), cv.mul(P(OPENING_AT_Z_OMEGA), rv))
        digest = cv.add(digest, cv.mul(P(GRAND_PRODUCT), rv))
        claims = (claims + zwz * rv) % q
        claims_com = cv.mul(vk.g1, claims)
        digest = cv.add(digest, cv.neg(claims_com))
        points_quotient = cv.mul(P(BATCH_OPENING_AT_Z), zeta)
        zeta_omega = zeta * vk.generator % q
        points_quotient = cv.add(points_quotient, cv.mul(P(OPENING_AT_Z_OMEGA), rv * zeta_omega % q))
        digest = cv.add(digest, points_quotient)
        T.update(digest=rb_ec(cv, digest), claims=claims, quotient=rb_ec(cv, cv.neg(quotient)))
        # :346-355  e(digest, G2_0) * e(-quotient, G2_1) == 1   <=>   digest == tau * quotient
        if vk.g2 is not None:
            # the reference's own final line: ec.pairing_check(EC.BLS12_381g1, digest + invert(quotient), g2)
            # (templateLogicSigBLS12_381.go:364-371)
            if cv.name == "bls12-381":
                from .pairing_bls12381 import pairing_check
            else:
                from .pairing_bn254 import pairing_check   # templateLogicSigBN254.go:350-355
            return pairing_check([digest, cv.neg(quotient)], [vk.g2[0], vk.g2[1]])
        if vk.tau is None:
            raise NotImplementedError("no tau and no G2 points: cannot finish the verification")
        return digest == cv.mul(quotient, vk.tau)
    except ValueError:
        return False


def rb_ec(cv: Curve, p: Point) -> bytes:
    """What the AVM ec ops return for a computed point: X||Y, all-zero for infinity."""
    if p is None:
        return bytes(2 * cv.fp_bytes)
    return cv.raw_bytes(p)
