"""ORACLE (test infrastructure): BN254 pairing check in plain Python, so the BN254 verifier transcription can end in the
reference's real `ec.pairing_check(EC.BN254g1, digest + quotient, g2)` (/root/reference/verifier/templateLogicSigBN254.go:350-355)
and not only in the known-tau shortcut.

Same textbook construction as oracle/pairing_bls12381.py with BN254's parameters: Fp12 = Fp[w]/(w^12 - 18 w^6 + 82),
Fp2 = Fp[i]/(i^2+1) embedded through i = w^6 - 9; the twist y^2 = x^3 + 3/(9+i) is mapped to E(Fp12) by
(x, y) -> (x * w^2, y * w^3); optimal-ate Miller loop over 6x+2 = 29793968203157093288 plus the two Frobenius line
evaluations, then the plain power (p^12 - 1)/r.

G2 encodings follow gnark: compressed = X.A1 || X.A0 big-endian, flags in the top TWO bits of byte 0
(10 = smaller y, 11 = larger y, 01 = infinity); setup/trusted_setup_test.go:33-39 pins vk.bin's first G2 point to the generator.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

P = 21888242871839275222246405745257275088696311157297823662689037894645226208583
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
ATE_LOOP = 29793968203157093288  # 6x + 2
MODULUS_COEFFS = (82, 0, 0, 0, 0, 0, -18, 0, 0, 0, 0, 0)  # w^12 = 18 w^6 - 82


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
        for top in range(22, 11, -1):       # reduce with w^12 = 18 w^6 - 82
            t = b[top]
            if t:
                b[top - 6] += 18 * t
                b[top - 12] -= 82 * t
        return Fq12(b[:12])

    def inv(self):
        """Extended Euclid on polynomials over Fp (modulus is irreducible)."""
        lm, hm = [1] + [0] * 12, [0] * 13
        low, high = self.c + [0], [c % P for c in MODULUS_COEFFS] + [1]     # w^12 - 18 w^6 + 82

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
W2 = W * W
W3 = W2 * W

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


B2 = f2_mul((3, 0), f2_inv((9, 1)))   # 3 / (9 + i)
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
    """gnark compressed G2: X.A1 || X.A0 (32 bytes each, big-endian), flags in the top two bits of byte 0."""
    assert len(b) == 64
    flags = b[0] & 0xC0
    if flags == 0x40:
        return None
    assert flags in (0x80, 0xC0), "not a compressed G2 point"
    a1 = int.from_bytes(bytes([b[0] & 0x3F]) + b[1:32], "big")
    a0 = int.from_bytes(b[32:], "big")
    x = (a0, a1)
    y = f2_sqrt(f2_add(f2_mul(f2_mul(x, x), x), B2))
    assert y is not None, "x not on the twist"
    largest = (y[1] > (P - 1) // 2) if y[1] else (y[0] > (P - 1) // 2)   # lexicographic: A1 first, then A0
    if largest != (flags == 0xC0):
        y = ((-y[0]) % P, (-y[1]) % P)
    return (x, y)


G2_GEN: G2Point = (
    (10857046999023057135944570762232829481370756359578518086990519993285655852781,
     11559732032986387107991004021392285783925812861821192530917403151452391805634),
    (8495653923123431417604973247489272438418190587263600148770280649306958101930,
     4082367875863433681332203403145435568316851327593401208105741076214120093531),
)

# ---- pairing -----------------------------------------------------------------------------------------------------


def _embed(a: Fq2) -> Fq12:
    """a0 + a1*i with i = w^6 - 9."""
    c = [0] * 12
    c[0] = a[0] - 9 * a[1]
    c[6] = a[1]
    return Fq12(c)


def _twist(Q: G2Point):
    x, y = Q
    return (_embed(x) * W2, _embed(y) * W3)


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
    q1 = (q[0] ** P, q[1] ** P)              # Frobenius
    nq2 = (q1[0] ** P, -(q1[1] ** P))        # -Frobenius^2
    f = f * _line(r, q1, p)
    r = _add(r, q1)
    f = f * _line(r, nq2, p)
    return f


def pairing_check(g1_points: List, g2_points: List[G2Point]) -> bool:
    """prod_i e(P_i, Q_i) == 1  (the AVM's ec.pairing_check)."""
    f = Fq12.one()
    for Pt, Q in zip(g1_points, g2_points):
        f = f * miller_loop(Q, Pt)
    return f ** ((P ** 12 - 1) // R) == Fq12.one()
