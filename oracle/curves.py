"""ORACLE (test infrastructure, not product code): curve + field arithmetic in plain Python ints.

This file is part of the CPU restatement ("oracle") of the PLONK prover path that AlgoPlonk reaches
through `plonk.Prove` (/root/reference/algoplonk.go:89).  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import it.  The arithmetic lives upstream in gnark-crypto v0.20.1
(go.mod:9, NOT vendored, not on this machine), so what is restated here is the published
mathematics; the constants are the ones the reference itself pins:

  * R_MOD / P_MOD            verifier/templateLogicSigBN254.go:15,18
                             verifier/templateLogicSigBLS12_381.go:15,18
  * G1 generators            setup/trusted_setup_test.go:33-36 (BN254 (1,2), observed in vk.bin),
                             setup/trusted_setup_test.go:54 (BLS12-381 generator hex)
  * compressed point flags   setup/trusted_setup_test.go:53-59,183-189 (KAT hex strings),
                             verifier/verifier.go:95-99 (0x40 = infinity in raw encoding)

PARITY STATUS: pinned by the reference's SRS decompression KATs (tests/test_oracle_kat.py) and, through the
executed verifier templates (tests/test_template_pin.py), for every group operation and encoding the
verification of a proof touches - including the encodings of the point at infinity (raw_bytes below).
MSM / NTT outputs against gnark's: no golden values exist in the reference.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple

Point = Optional[Tuple[int, int]]  # affine (x, y); None = point at infinity


@dataclass(frozen=True)
class Curve:
    name: str
    curve_id: int          # C-ABI id: 0 = BN254, 1 = BLS12-381 (include/apk.h)
    r: int                 # scalar field modulus
    p: int                 # base field modulus
    b: int                 # y^2 = x^3 + b
    g1: Tuple[int, int]
    fr_root: int           # primitive 2^fr_adicity-th root of unity in Fr  [UPSTREAM gnark-crypto fr/fft]
    fr_adicity: int
    coset_shift: int       # gnark fr.MultiplicativeGen; surfaced as VK CosetShift (templateLogicSigBN254.go:68)
    fp_bytes: int          # 32 / 48
    fr_bytes: int = 32

    # ---- Fr -------------------------------------------------------------------------------
    def fr_inv(self, a: int) -> int:
        return pow(a, -1, self.r)

    def omega(self, n: int) -> int:
        """Generator of the size-n (power of two) subgroup, as gnark's fft.NewDomain picks it
        (VK `Generator`, templateLogicSigBN254.go:57)."""
        k = n.bit_length() - 1
        assert 1 << k == n and k <= self.fr_adicity
        return pow(self.fr_root, 1 << (self.fr_adicity - k), self.r)

    # ---- G1 (affine, plain ints) -----------------------------------------------------------
    def is_on_curve(self, P: Point) -> bool:
        if P is None:
            return True
        x, y = P
        return (y * y - x * x * x - self.b) % self.p == 0

    def neg(self, P: Point) -> Point:
        if P is None:
            return None
        return (P[0], (-P[1]) % self.p)

    def add(self, P: Point, Q: Point) -> Point:
        p = self.p
        if P is None:
            return Q
        if Q is None:
            return P
        x1, y1 = P
        x2, y2 = Q
        if x1 == x2:
            if (y1 + y2) % p == 0:
                return None
            lam = 3 * x1 * x1 * pow(2 * y1, -1, p) % p
        else:
            lam = (y2 - y1) * pow(x2 - x1, -1, p) % p
        x3 = (lam * lam - x1 - x2) % p
        return (x3, (lam * (x1 - x3) - y1) % p)

    def sub(self, P: Point, Q: Point) -> Point:
        return self.add(P, self.neg(Q))

    # Jacobian internals for a faster scalar multiplication
    def _jdbl(self, X, Y, Z):
        p = self.p
        if Z == 0 or Y == 0:
            return (1, 1, 0)
        A = X * X % p
        B = Y * Y % p
        C = B * B % p
        D = 2 * ((X + B) * (X + B) - A - C) % p
        E = 3 * A % p
        X3 = (E * E - 2 * D) % p
        Y3 = (E * (D - X3) - 8 * C) % p
        Z3 = 2 * Y * Z % p
        return (X3, Y3, Z3)

    def _jadd_affine(self, X1, Y1, Z1, x2, y2):
        p = self.p
        if Z1 == 0:
            return (x2, y2, 1)
        Z1Z1 = Z1 * Z1 % p
        U2 = x2 * Z1Z1 % p
        S2 = y2 * Z1 * Z1Z1 % p
        if U2 == X1:
            if S2 == Y1:
                return self._jdbl(X1, Y1, Z1)
            return (1, 1, 0)
        H = (U2 - X1) % p
        R = (S2 - Y1) % p
        HH = H * H % p
        HHH = H * HH % p
        V = X1 * HH % p
        X3 = (R * R - HHH - 2 * V) % p
        Y3 = (R * (V - X3) - Y1 * HHH) % p
        Z3 = Z1 * H % p
        return (X3, Y3, Z3)

    def _jaffine(self, X, Y, Z) -> Point:
        if Z == 0:
            return None
        p = self.p
        zi = pow(Z, -1, p)
        zi2 = zi * zi % p
        return (X * zi2 % p, Y * zi2 * zi % p)

    def mul(self, P: Point, k: int) -> Point:
        k %= self.r
        if P is None or k == 0:
            return None
        x, y = P
        X, Y, Z = 1, 1, 0
        for bit in bin(k)[2:]:
            X, Y, Z = self._jdbl(X, Y, Z)
            if bit == "1":
                X, Y, Z = self._jadd_affine(X, Y, Z, x, y)
        return self._jaffine(X, Y, Z)

    def msm_naive(self, points, scalars) -> Point:
        """Sum s_i * P_i, one double-and-add per term.  The definition every MSM must equal
        (gnark-crypto G1Affine.MultiExp [UPSTREAM], reached from algoplonk.go:89)."""
        acc: Point = None
        for P, s in zip(points, scalars):
            acc = self.add(acc, self.mul(P, s))
        return acc

    # ---- encodings -------------------------------------------------------------------------
    def raw_bytes(self, P: Point) -> bytes:
        """gnark `RawBytes()` / `Marshal()`: X || Y big-endian.  Infinity: BLS12-381 = 0x40 then zeros
        (helper.go:35-72 uses RawBytes; verifier/verifier.go:95-99 documents 0x40 and keeps it for the
        transcript through `hexEncoded`, :102-105); BN254 = all zeros: its template has no `_fs` constants,
        the bytes that are hashed are the bytes the AVM's ec ops decode (templateLogicSigBN254.go:57-61,
        131-132), and those accept only the all-zero encoding.  Pinned by the executed template
        (tests/golden/template_verdicts.json: pythagorean has [Qk] = infinity, identity has [Qm] = infinity)."""
        n = self.fp_bytes
        if P is None:
            return bytes([0x40 if n == 48 else 0x00]) + bytes(2 * n - 1)
        return P[0].to_bytes(n, "big") + P[1].to_bytes(n, "big")

    def from_raw_bytes(self, b: bytes) -> Point:
        n = self.fp_bytes
        assert len(b) == 2 * n
        if not any(b[1:]) and b[0] in (0x00, 0x40):
            return None
        return (int.from_bytes(b[:n], "big"), int.from_bytes(b[n:], "big"))

    def _lex_largest(self, y: int) -> bool:
        return y > (self.p - 1) // 2

    def compress(self, P: Point) -> bytes:
        """gnark compressed G1 encoding (SRS pk.bin / vk.bin)."""
        n = self.fp_bytes
        if self.name == "bls12-381":
            if P is None:
                return bytes([0xC0]) + bytes(n - 1)
            flag = 0xA0 if self._lex_largest(P[1]) else 0x80
        else:
            if P is None:
                return bytes([0x40]) + bytes(n - 1)
            flag = 0xC0 if self._lex_largest(P[1]) else 0x80
        out = bytearray(P[0].to_bytes(n, "big"))
        out[0] |= flag
        return bytes(out)

    def decompress(self, b: bytes) -> Point:
        """Inverse of `compress`: one Fp square root per point (setup/setup.go:173-174,189-190 ->
        kzg SRS ReadFrom [UPSTREAM]).  KATs: setup/trusted_setup_test.go:53-59,183-189,256."""
        n = self.fp_bytes
        assert len(b) == n
        if self.name == "bls12-381":
            flags = b[0] & 0xE0
            if flags == 0xC0:
                return None
            assert flags in (0x80, 0xA0), "not a compressed BLS12-381 G1 point"
            largest = flags == 0xA0
            x = int.from_bytes(bytes([b[0] & 0x1F]) + b[1:], "big")
        else:
            flags = b[0] & 0xC0
            if flags == 0x40:
                return None
            assert flags in (0x80, 0xC0), "not a compressed BN254 G1 point"
            largest = flags == 0xC0
            x = int.from_bytes(bytes([b[0] & 0x3F]) + b[1:], "big")
        assert x < self.p
        y = sqrt_mod(( x * x * x + self.b) % self.p, self.p)
        assert y is not None, "x not on curve"
        if self._lex_largest(y) != largest:
            y = self.p - y
        return (x, y)


def sqrt_mod(a: int, p: int) -> Optional[int]:
    """Square root mod p for p = 3 mod 4 (true for both base fields)."""
    assert p % 4 == 3
    y = pow(a, (p + 1) // 4, p)
    return y if y * y % p == a % p else None


BN254 = Curve(
    name="bn254",
    curve_id=0,
    r=21888242871839275222246405745257275088548364400416034343698204186575808495617,
    p=21888242871839275222246405745257275088696311157297823662689037894645226208583,
    b=3,
    g1=(1, 2),
    fr_root=19103219067921713944291392827692070036145651957329286315305642004821462161904,
    fr_adicity=28,
    coset_shift=5,
    fp_bytes=32,
)

BLS12_381 = Curve(
    name="bls12-381",
    curve_id=1,
    r=52435875175126190479447740508185965837690552500527637822603658699938581184513,
    p=4002409555221667393417789825735904156556882819939007885332058136124031650490837864442687629129015664037894272559787,
    b=4,
    g1=(
        0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB,
        0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1,
    ),
    fr_root=10238227357739495823651030575849232062558860180284477541189508159991286009131,
    fr_adicity=32,
    coset_shift=7,
    fp_bytes=48,
)

CURVES = {"bn254": BN254, "bls12-381": BLS12_381, 0: BN254, 1: BLS12_381}
