"""ORACLE (test infrastructure): BLS12-381 pairing check in plain Python, so the verifier transcription can end in the
reference's real `ec.pairing_check(EC.BLS12_381g1, digest + quotient, g2)`
(/root/reference/verifier/templateLogicSigBLS12_381.go:366-371) instead of the known-tau shortcut - needed for proofs
made under the real Ethereum KZG ceremony SRS, where nobody knows tau.

Textbook construction (slow, a few seconds per check; only used on a handful of test vectors):
Fp12 = Fp[w]/(w^12 - 2 w^6 + 2) with Fp2 = Fp[i]/(i^2+1) embedded through i = w^6 - 1; G2 lives on the twist
y^2 = x^3 + 4(1+i) and is mapped to E(Fp12) by (x, y) -> (x / w^2, y / w^3); Miller loop over |x| = 0xd201000000010000
with affine line functions, then the plain power (p^12 - 1)/r.  The sign of x only inverts the pairing value, which a
product-equals-one check does not see.

G2 encodings follow gnark: compressed = X.A1 || X.A0 big-endian with the flags in byte 0
(setup/trusted_setup_test.go:93-96,112-113).
"""
from __future__ import annotations

from typing import List, Optional, Tuple

P = 4002409555221667393417789825735904156556882819939007885332058136124031650490837864442687629129015664037894272559787
R = 52435875175126190479447740508185965837690552500527637822603658699938581184513
ATE_LOOP = 0xD201000000010000
MODULUS_COEFFS = (2, 0, 0, 0, 0, 0, -2, 0, 0, 0, 0, 0)  # w^12 = 2 w^6 - 2


class Fq12:
    __slots__ = ("c",)

    def __init__(self, c):
        self.c = [x % P for x in c]

    @staticmethod
    def one():
        return Fq12([1] + [0] * 11)

    @staticmethod
    def zero():
        return Fq12([0] * 12)

    def __add__(self, o):
        return Fq12([a + b for a, b in zip(self.c, o.c)])

    def __sub__(self, o):
        return Fq12([a - b for a, b in zip(self.c, o.c)])

    def __neg__(self):
        return Fq12([-a for a in self.c])

    def __eq__(self, o):
        return self.c == o.c

    def scale(self, k: int):
        return Fq12([a * k for a in self.c])

    def __mul__(self, o):
        b = [0] * 23
        for i, x in enumerate(self.c):
            if x:
                for j, y in enumerate(o.c):
                    b[i + j] += x * y
        for top in range(22, 11, -1):       # reduce with w^12 = 2 w^6 - 2
            t = b[top]
            if t:
                b[top - 6] += 2 * t
                b[top - 12] -= 2 * t
        return Fq12(b[:12])

    def inv(self):
        """Extended Euclid on polynomials over Fp (modulus is irreducible)."""
        lm, hm = [1] + [0] * 12, [0] * 13
        low, high = self.c + [0], [c % P for c in MODULUS_COEFFS] + [1]     # w^12 - 2 w^6 + 2

        def deg(p):
            d = len(p) - 1
            while d and p[d] == 0:
                d -= 1
            return d

        def poly_rounded_div(a, b):
            dega, degb = deg(a), deg(b)
            temp = list(a)
            o = [0] * len(a)
            for i in range(dega - degb, -1, -1):
                o[i] = (o[i] + temp[degb + i] * pow(b[degb], -1, P)) % P
                for c in range(degb + 1):
                    temp[c + i] = (temp[c + i] - o[i] * b[c]) % P
            return o[: deg(o) + 1]

        while deg(low):
            r = poly_rounded_div(high, low)
            r += [0] * (13 - len(r))
            nm, new = list(hm), list(high)
            for i in range(13):
                for j in range(13 - i):
                    nm[i + j] = (nm[i + j] - lm[i] * r[j]) % P
                    new[i + j] = (new[i + j] - low[i] * r[j]) % P
            lm, low, hm, high = nm, new, lm, low
        k = pow(low[0], -1, P)
        return Fq12([x * k for x in lm[:12]])

    def __truediv__(self, o):
        return self * o.inv()

    def __pow__(self, e: int):
        r, b = Fq12.one(), self
        while e:
            if e & 1:
                r = r * b
            b = b * b
            e >>= 1
        return r


W = Fq12([0, 1] + [0] * 10)
W2_INV = (W * W).inv()
W3_INV = (W * W * W).inv()

# ---- Fp2 as pairs (a0, a1) = a0 + a1*i ------------------------------------------------------------------------------
Fq2 = Tuple[int, int]


def f2_mul(a: Fq2, b: Fq2) -> Fq2:
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def f2_add(a, b):
    return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)


def f2_sub(a, b):
    return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)


def f2_inv(a):
    d = pow(a[0] * a[0] + a[1] * a[1], -1, P)
    return (a[0] * d % P, -a[1] * d % P)


def f2_pow(a, e):
    r = (1, 0)
    while e:
        if e & 1:
            r = f2_mul(r, a)
        a = f2_mul(a, a)
        e >>= 1
    return r


def f2_sqrt(a: Fq2) -> Optional[Fq2]:
    """p = 3 mod 4 (Adj-Rodriguez-Henriquez)."""
    if a == (0, 0):
        return (0, 0)
    a1 = f2_pow(a, (P - 3) // 4)
    alpha = f2_mul(f2_mul(a1, a1), a)
    x0 = f2_mul(a1, a)
    if alpha == (P - 1, 0):
        x = f2_mul((0, 1), x0)
    else:
        b = f2_pow(f2_add((1, 0), alpha), (P - 1) // 2)
        x = f2_mul(b, x0)
    return x if f2_mul(x, x) == (a[0] % P, a[1] % P) else None


B2 = (4, 4)
G2Point = Optional[Tuple[Fq2, Fq2]]


def g2_on_curve(Q: G2Point) -> bool:
    if Q is None:
        return True
    x, y = Q
    return f2_sub(f2_mul(y, y), f2_add(f2_mul(f2_mul(x, x), x), B2)) == (0, 0)


def g2_add(A: G2Point, Bp: G2Point) -> G2Point:
    if A is None:
        return Bp
    if Bp is None:
        return A
    (x1, y1), (x2, y2) = A, Bp
    if x1 == x2:
        if f2_add(y1, y2) == (0, 0):
            return None
        m = f2_mul(f2_mul((3, 0), f2_mul(x1, x1)), f2_inv(f2_mul((2, 0), y1)))
    else:
        m = f2_mul(f2_sub(y2, y1), f2_inv(f2_sub(x2, x1)))
    x3 = f2_sub(f2_sub(f2_mul(m, m), x1), x2)
    return (x3, f2_sub(f2_mul(m, f2_sub(x1, x3)), y1))


def g2_mul(Q: G2Point, k: int) -> G2Point:
    acc = None
    k %= R
    for bit in bin(k)[2:] if k else "":
        acc = g2_add(acc, acc)
        if bit == "1":
            acc = g2_add(acc, Q)
    return acc


def g2_decompress(b: bytes) -> G2Point:
    """gnark compressed G2: X.A1 || X.A0 (48 bytes each, big-endian), flags in the top three bits of byte 0."""
    assert len(b) == 96
    flags = b[0] & 0xE0
    if flags == 0xC0:
        return None
    assert flags in (0x80, 0xA0), "not a compressed G2 point"
    a1 = int.from_bytes(bytes([b[0] & 0x1F]) + b[1:48], "big")
    a0 = int.from_bytes(b[48:], "big")
    x = (a0, a1)
    y = f2_sqrt(f2_add(f2_mul(f2_mul(x, x), x), B2))
    assert y is not None, "x not on the twist"
    largest = (y[1] > (P - 1) // 2) if y[1] else (y[0] > (P - 1) // 2)   # lexicographic: A1 first, then A0
    if largest != (flags == 0xA0):
        y = ((-y[0]) % P, (-y[1]) % P)
    return (x, y)


G2_GEN: G2Point = (
    (0x024AA2B2F08F0A91260805272DC51051C6E47AD4FA403B02B4510B647AE3D1770BAC0326A805BBEFD48056C8C121BDB8,
     0x13E02B6052719F607DACD3A088274F65596BD0D09920B61AB5DA61BBDC7F5049334CF11213945D57E5AC7D055D042B7E),
    (0x0CE5D527727D6E118CC9CDC6DA2E351AADFD9BAA8CBDD3A76D429A695160D12C923AC9CC3BACA289E193548608B82801,
     0x0606C4A02EA734CC32ACD2B02BC28B99CB3E287E85A763AF267492AB572E99AB3F370D275CEC1DA1AAA9075FF05F79BE),
)

# ---- pairing -----------------------------------------------------------------------------------------------------


def _embed(a: Fq2) -> Fq12:
    """a0 + a1*i with i = w^6 - 1."""
    c = [0] * 12
    c[0] = a[0] - a[1]
    c[6] = a[1]
    return Fq12(c)


def _twist(Q: G2Point):
    x, y = Q
    return (_embed(x) * W2_INV, _embed(y) * W3_INV)


def _cast(Pt):
    return (Fq12([Pt[0]] + [0] * 11), Fq12([Pt[1]] + [0] * 11))


def _double(pt):
    x, y = pt
    m = (x * x).scale(3) / y.scale(2)
    nx = m * m - x.scale(2)
    return (nx, m * (x - nx) - y)


def _add(p1, p2):
    if p1 is None:
        return p2
    if p2 is None:
        return p1
    x1, y1 = p1
    x2, y2 = p2
    if x1 == x2 and y1 == y2:
        return _double(p1)
    if x1 == x2:
        return None
    m = (y2 - y1) / (x2 - x1)
    nx = m * m - x1 - x2
    return (nx, m * (x1 - nx) - y1)


def _line(p1, p2, t):
    x1, y1 = p1
    x2, y2 = p2
    xt, yt = t
    if not (x1 == x2):
        m = (y2 - y1) / (x2 - x1)
        return m * (xt - x1) - (yt - y1)
    if y1 == y2:
        m = (x1 * x1).scale(3) / y1.scale(2)
        return m * (xt - x1) - (yt - y1)
    return xt - x1


def miller_loop(Q: G2Point, Pt) -> Fq12:
    if Q is None or Pt is None:
        return Fq12.one()
    q, p = _twist(Q), _cast(Pt)
    r, f = q, Fq12.one()
    for i in range(ATE_LOOP.bit_length() - 2, -1, -1):
        f = f * f * _line(r, r, p)
        r = _double(r)
        if (ATE_LOOP >> i) & 1:
            f = f * _line(r, q, p)
            r = _add(r, q)
    return f


def pairing_check(g1_points: List, g2_points: List[G2Point]) -> bool:
    """prod_i e(P_i, Q_i) == 1  (the AVM's ec.pairing_check)."""
    f = Fq12.one()
    for Pt, Q in zip(g1_points, g2_points):
        f = f * miller_loop(Q, Pt)
    return f ** ((P ** 12 - 1) // R) == Fq12.one()
