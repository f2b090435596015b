"""Setup registry and SRS handling: the Python mirror of /root/reference/setup/setup.go.

* `Name` / `Get` / registry            setup/setup.go:23-36,49-86
* `Run(ccs, name)`                     setup/setup.go:95-150  -> here: SRS -> libapk context (plonk.Setup's
                                        8+k commitments run on the GPU inside apk_ctx_create) -> (pk, vk)
* TestOnly setups                      setup/setup.go:102-108 (gnark test/unsafekzg: SRS from a random tau);
                                        the tau^i * G1 scalar multiplications run on the GPU (apk_g1_mul_batch)
* trusted setups                       setup/setup.go:110-149,196-228: pk.bin = BE u32 count || compressed G1.
                                        The ceremony files are not shipped with this repo; point
                                        APK_TRUSTED_SETUP_DIR at a directory laid out like the reference's
                                        `setup/<NamePath>/{pk,vk}.bin`.  Point decompression and `ToLagrangeG1` run on
                                        the GPU (apk_g1_decompress / apk_g1_to_lagrange); the Lagrange
                                        SRS is only built for BSB22 circuits (wire commitments use the canonical SRS).
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from enum import IntEnum
from typing import Dict, List, Optional, Tuple

from . import ecc
from ._lib import lib, check


class Name(IntEnum):
    """setup.Name (setup/setup.go:23-36): the numeric values are the reference's iota order, so an integer id carried
    over from the Go side names the same setup (0 is the trusted PPoT ceremony, never a TestOnly setup)."""
    PerpetualPowersOfTauBN254 = 0
    EthereumKzgCeremonyBLS12381 = 1
    DuskBLS12381 = 2
    TestOnlyBN254 = 3
    TestOnlyBLS12381 = 4


@dataclass(frozen=True)
class Setup:
    Curve: ecc.ID
    Trusted: bool
    NamePath: str
    MaxConstraints: int


_setups: Dict[Name, Setup] = {
    Name.PerpetualPowersOfTauBN254: Setup(ecc.BN254, True, "PerpetualPowersOfTauBN254", 1 << 17),
    Name.EthereumKzgCeremonyBLS12381: Setup(ecc.BLS12_381, True, "EethereumKzgCeremonyBLS12_381", 1 << 14),
    Name.DuskBLS12381: Setup(ecc.BLS12_381, True, "DuskBLS12_381", 1 << 20),
    Name.TestOnlyBN254: Setup(ecc.BN254, False, "test_only", 1 << 24),
    Name.TestOnlyBLS12381: Setup(ecc.BLS12_381, False, "test_only", 1 << 24),
}


def Get(name) -> Tuple[Optional[Setup], bool]:
    """setup.Get (setup/setup.go:78-86): (setup, ok)."""
    try:
        s = _setups.get(Name(name))
    except ValueError:
        s = None
    return s, s is not None


def TestOnlySetup(curve: ecc.ID) -> Name:
    """setup.TestOnlySetup (setup/setup.go:152-161)."""
    if curve is ecc.BLS12_381:
        return Name.TestOnlyBLS12381
    if curve is ecc.BN254:
        return Name.TestOnlyBN254
    raise ValueError("unsupported curve: %s" % curve)


# ---- SRS -------------------------------------------------------------------------------------------------

@dataclass
class SRS:
    """kzg.SRS in G1: canonical (n+3 points) and, when needed, Lagrange (n points); gnark in-memory bytes."""
    curve: ecc.ID
    n: int
    g1: bytes
    g1_lagrange: Optional[bytes]
    tau: Optional[int] = None   # TestOnly setups only: the toxic value (never known for a trusted setup)
    g2: Optional[bytes] = None  # kzg.VerifyingKey.G2 = ([1]G2, [tau]G2), gnark in-memory G2Affine (X.A0 || X.A1 || Y.A0 || Y.A1 each)


def _mul_base_batch(curve: ecc.ID, scalars: List[int], device: int) -> bytes:
    sc = curve.fr_vector(scalars)
    out = C.create_string_buffer(len(scalars) * 2 * curve.fp_bytes)
    base = curve.g1_to_bytes(curve.g1)
    check(lib.apk_g1_mul_batch(curve.abi, device, base, sc, len(scalars), out))
    return out.raw


def unsafe_srs(curve: ecc.ID, n: int, tau: int, device: int = 0, lagrange: bool = False) -> SRS:
    """unsafekzg.NewSRS analogue (setup/setup.go:103): powers of a KNOWN tau - never for production."""
    r = curve.r
    pw = [1] * (n + 3)
    for i in range(1, n + 3):
        pw[i] = pw[i - 1] * tau % r
    g1 = _mul_base_batch(curve, pw, device)
    lag = None
    if lagrange:
        # L_i(tau) = omega^i (tau^n - 1) / (n (tau - omega^i)), one batch inversion
        w = curve.omega(n)
        zn = (pow(tau, n, r) - 1) * pow(n, -1, r) % r
        wi, ws, den = 1, [], []
        for _ in range(n):
            ws.append(wi)
            den.append((tau - wi) % r)
            wi = wi * w % r
        pref = [1] * (n + 1)
        for i, d in enumerate(den):
            pref[i + 1] = pref[i] * d % r
        inv = pow(pref[n], -1, r)
        sc = [0] * n
        for i in range(n - 1, -1, -1):
            sc[i] = ws[i] * zn % r * (inv * pref[i] % r) % r
            inv = inv * den[i] % r
        lag = _mul_base_batch(curve, sc, device)
    return SRS(curve, n, g1, lag, tau, g2_from_tau(curve, tau))


def g2_from_tau(curve: ecc.ID, tau: int) -> bytes:
    """([1]G2, [tau]G2) for a TestOnly SRS (host: apk_g2_mul_generator)."""
    out = b""
    for k in (1, tau):
        buf = C.create_string_buffer(4 * curve.fp_bytes)
        check(lib.apk_g2_mul_generator(curve.abi, curve.fr_vector([k]), buf))
        out += buf.raw
    return out


def g2_from_vk_bin(curve: ecc.ID, vk: bytes) -> bytes:
    """The two G2 points of a trusted setup's vk.bin = G2[0] || G2[1] || G1[0], all compressed (writer
    order setup/DuskBLS12_381/audit.go:155-179), decompressed on the host (apk_g2_decompress)."""
    w = 2 * curve.fp_bytes
    if len(vk) != 2 * w + curve.fp_bytes:
        raise ValueError("vk.bin: expected %d bytes, found %d" % (2 * w + curve.fp_bytes, len(vk)))
    out = b""
    for j in range(2):
        buf = C.create_string_buffer(4 * curve.fp_bytes)
        check(lib.apk_g2_decompress(curve.abi, vk[j * w: (j + 1) * w], buf))
        out += buf.raw
    return out


def _sqrt_fp(a: int, p: int) -> Optional[int]:
    y = pow(a, (p + 1) // 4, p)   # both base fields are 3 mod 4
    return y if y * y % p == a % p else None


def decompress_g1(curve: ecc.ID, b: bytes) -> ecc.Point:
    """gnark compressed G1 (KATs /root/reference/setup/trusted_setup_test.go:53-59,183-189)."""
    n = curve.fp_bytes
    if curve is ecc.BLS12_381:
        flags = b[0] & 0xE0
        if flags == 0xC0:
            return None
        if flags not in (0x80, 0xA0):
            raise ValueError("not a compressed BLS12-381 G1 point")
        largest = flags == 0xA0
        x = int.from_bytes(bytes([b[0] & 0x1F]) + b[1:n], "big")
        bcoef = 4
    else:
        flags = b[0] & 0xC0
        if flags == 0x40:
            return None
        if flags not in (0x80, 0xC0):
            raise ValueError("not a compressed BN254 G1 point")
        largest = flags == 0xC0
        x = int.from_bytes(bytes([b[0] & 0x3F]) + b[1:n], "big")
        bcoef = 3
    y = _sqrt_fp((x * x * x + bcoef) % curve.p, curve.p)
    if y is None:
        raise ValueError("x is not on the curve")
    if (y > (curve.p - 1) // 2) != largest:
        y = curve.p - y
    return (x, y)


def load_trusted_setup_bytes(dirpath: str, g1_count: int, g1_compressed_size: int) -> Tuple[bytes, bytes]:
    """loadTrustedSetupBytes (setup/setup.go:196-228), same checks and the same truncation."""
    if g1_count < 2:
        raise ValueError("need at least 2 G1 points")
    with open(os.path.join(dirpath, "pk.bin"), "rb") as f:
        g1 = f.read(4 + g1_count * g1_compressed_size)
    with open(os.path.join(dirpath, "vk.bin"), "rb") as f:
        vk = f.read()
    needed = 4 + g1_count * g1_compressed_size
    declared = int.from_bytes(g1[:4], "big")
    if len(g1) < needed or declared < g1_count:
        raise ValueError("pk.bin too small for %d elements" % g1_count)
    return g1_count.to_bytes(4, "big") + g1[4:needed], vk


def decompress_g1_batch(curve: ecc.ID, compressed: bytes, device: int = 0) -> bytes:
    """kzg SRS ReadFrom on the GPU (apk_g1_decompress): compressed points -> gnark in-memory affine bytes."""
    count = len(compressed) // curve.fp_bytes
    out = C.create_string_buffer(count * 2 * curve.fp_bytes)
    check(lib.apk_g1_decompress(curve.abi, device, compressed, count, out))
    return out.raw


def to_lagrange_g1(curve: ecc.ID, points: bytes, device: int = 0) -> bytes:
    """kzg.ToLagrangeG1 on the GPU (apk_g1_to_lagrange): [tau^j]G1, j < n  ->  [L_i(tau)]G1."""
    n = len(points) // (2 * curve.fp_bytes)
    out = C.create_string_buffer(len(points))
    check(lib.apk_g1_to_lagrange(curve.abi, device, points, n, out))
    return out.raw


def trusted_srs(setup: Setup, n: int, device: int = 0, lagrange: bool = False, root: Optional[str] = None) -> SRS:
    """trustedSetupBLS12381 / trustedSetupBN254 + the ToLagrangeG1 step of setup.Run (setup/setup.go:110-143)."""
    root = root or os.environ.get("APK_TRUSTED_SETUP_DIR")
    if not root:
        raise FileNotFoundError("trusted setup files are not bundled: set APK_TRUSTED_SETUP_DIR")
    cv = setup.Curve
    g1b, vk = load_trusted_setup_bytes(os.path.join(root, setup.NamePath), n + 3, cv.fp_bytes)
    g1 = decompress_g1_batch(cv, g1b[4:], device)
    lag = to_lagrange_g1(cv, g1[: n * 2 * cv.fp_bytes], device) if lagrange else None   # G1[:len-3] (setup.go:124,138)
    g2 = g2_from_vk_bin(cv, vk)                                                          # srs.Vk.ReadFrom (setup.go:174,190)
    if decompress_g1_batch(cv, vk[-cv.fp_bytes:], device) != g1[: 2 * cv.fp_bytes]:
        raise ValueError("vk.bin: Vk.G1 differs from pk.bin's first point")               # setup/trusted_setup_test.go:127-129
    return SRS(cv, n, g1, lag, None, g2)
