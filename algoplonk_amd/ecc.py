"""Curve ids and the gnark in-memory encodings the C-ABI speaks (include/apk.h): Fr/Fp elements are
little-endian Montgomery limbs, G1 affine is X||Y with (0,0) as infinity.

Mirrors `ecc.ID` as the reference uses it (`ecc.BN254`, `ecc.BLS12_381`: /root/reference/algoplonk.go:39-41);
moduli are the reference's constants (/root/reference/verifier/templateLogicSigBN254.go:15,18 and
templateLogicSigBLS12_381.go:15,18).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Iterable, List, Optional, Sequence, Tuple

Point = Optional[Tuple[int, int]]


@dataclass(frozen=True)
class ID:
    name: str
    abi: int          # APK_BN254 / APK_BLS12_381
    r: int            # scalar field
    p: int            # base field
    fp_bytes: int
    g1: Tuple[int, int]
    fr_root: int      # 2^adicity-th root of unity
    fr_adicity: int
    coset_shift: int

    def __str__(self) -> str:
        return self.name

    @property
    def fp_R(self) -> int:
        return 1 << (8 * self.fp_bytes)

    def ScalarField(self) -> int:
        return self.r

    # ---- Fr ----
    def fr_to_mont_bytes(self, x: int) -> bytes:
        return ((x % self.r) * (1 << 256) % self.r).to_bytes(32, "little")

    def fr_from_mont_bytes(self, b: bytes) -> int:
        return int.from_bytes(b, "little") * pow(1 << 256, -1, self.r) % self.r

    def fr_vector(self, xs: Iterable[int]) -> bytes:
        r = self.r
        return b"".join(((x % r) * (1 << 256) % r).to_bytes(32, "little") for x in xs)

    def fr_vector_decode(self, b: bytes) -> List[int]:
        rinv = pow(1 << 256, -1, self.r)
        return [int.from_bytes(b[i:i + 32], "little") * rinv % self.r for i in range(0, len(b), 32)]

    # ---- G1 ----
    def g1_to_bytes(self, P: Point) -> bytes:
        n = self.fp_bytes
        if P is None:
            return bytes(2 * n)
        R = self.fp_R
        return (P[0] * R % self.p).to_bytes(n, "little") + (P[1] * R % self.p).to_bytes(n, "little")

    def g1_from_bytes(self, b: bytes) -> Point:
        n = self.fp_bytes
        b = bytes(b[: 2 * n])
        if not any(b):
            return None
        rinv = pow(self.fp_R, -1, self.p)
        return (int.from_bytes(b[:n], "little") * rinv % self.p, int.from_bytes(b[n:], "little") * rinv % self.p)

    def g1_vector(self, pts: Sequence[Point]) -> bytes:
        return b"".join(self.g1_to_bytes(P) for P in pts)

    def g1_vector_decode(self, b: bytes) -> List[Point]:
        m = 2 * self.fp_bytes
        return [self.g1_from_bytes(b[i:i + m]) for i in range(0, len(b), m)]

    def omega(self, n: int) -> int:
        k = n.bit_length() - 1
        assert 1 << k == n and k <= self.fr_adicity
        return pow(self.fr_root, 1 << (self.fr_adicity - k), self.r)


BN254 = ID(
    name="bn254", abi=0,
    r=21888242871839275222246405745257275088548364400416034343698204186575808495617,
    p=21888242871839275222246405745257275088696311157297823662689037894645226208583,
    fp_bytes=32, g1=(1, 2),
    fr_root=19103219067921713944291392827692070036145651957329286315305642004821462161904, fr_adicity=28,
    coset_shift=5,
)

BLS12_381 = ID(
    name="bls12_381", abi=1,
    r=52435875175126190479447740508185965837690552500527637822603658699938581184513,
    p=4002409555221667393417789825735904156556882819939007885332058136124031650490837864442687629129015664037894272559787,
    fp_bytes=48,
    g1=(0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB,
        0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1),
    fr_root=10238227357739495823651030575849232062558860180284477541189508159991286009131, fr_adicity=32,
    coset_shift=7,
)

# curves the AVM does not support, present so `Compile` can reject them like algoplonk.go:39-41 does
BLS12_377 = ID(name="bls12_377", abi=-1, r=0, p=0, fp_bytes=0, g1=(0, 0), fr_root=0, fr_adicity=0, coset_shift=0)
