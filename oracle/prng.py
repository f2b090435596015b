"""ORACLE (test infrastructure): the seeded generator every synthetic workload is drawn from.

BASELINE.md §2: "scalars uniform in Fr via SplitMix64 + rejection; SRS from tau = SHA-256(seed) mod r".
The same generator is restated in C (oracle/apk_oracle.c) and in the product's bench harness so the
CPU baseline, the HIP path and the golden fixtures all see identical inputs.
"""
from __future__ import annotations

import hashlib

MASK64 = (1 << 64) - 1


class SplitMix64:
    def __init__(self, seed: int):
        self.s = seed & MASK64

    def next(self) -> int:
        self.s = (self.s + 0x9E3779B97F4A7C15) & MASK64
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
        return z ^ (z >> 31)

    def fr(self, r: int) -> int:
        """Uniform element of [0, r): four 64-bit words little-endian, top bits masked to the
        bit length of r, rejected if >= r."""
        bits = r.bit_length()
        while True:
            v = 0
            for i in range(4):
                v |= self.next() << (64 * i)
            v &= (1 << bits) - 1
            if v < r:
                return v

    def below(self, n: int) -> int:
        """Uniform integer in [0, n) for n < 2^63 (rejection on the top multiple)."""
        lim = (1 << 64) - ((1 << 64) % n)
        while True:
            v = self.next()
            if v < lim:
                return v % n


def tau_from_seed(seed: int, r: int) -> int:
    return int.from_bytes(hashlib.sha256(seed.to_bytes(8, "big")).digest(), "big") % r
