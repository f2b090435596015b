"""ORACLE (test infrastructure): ctypes loader for oracle/liboracle.so (the plain-C restatement, apk_oracle.c).
Same gnark in-memory layouts as include/apk.h, so tests feed identical buffers to both sides."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "liboracle.so")


class Circuit(C.Structure):
    _fields_ = [("curve", C.c_int), ("n", C.c_uint64), ("nb_public", C.c_uint32), ("srs", C.c_void_p),
                ("ql", C.c_void_p), ("qr", C.c_void_p), ("qm", C.c_void_p), ("qo", C.c_void_p), ("qk", C.c_void_p),
                ("perm", C.c_void_p)]


class CircuitEx(C.Structure):
    """orc_circuit_ex: orc_circuit + the BSB22 part of the trace (Qcp columns, commitment constraint indexes) and the Lagrange SRS."""
    _fields_ = [("base", Circuit), ("nb_commit", C.c_uint32), ("cci", C.c_uint32 * 2), ("qcp", C.c_void_p * 2), ("srs_lagrange", C.c_void_p)]


def load() -> C.CDLL:
    if not os.path.exists(_PATH):
        subprocess.check_call(["make", "-C", _HERE])
    lib = C.CDLL(_PATH)
    vp = C.c_void_p
    lib.orc_msm.argtypes = [C.c_int, vp, vp, C.c_uint64, C.c_int, vp]
    lib.orc_ntt.argtypes = [C.c_int, vp, C.c_uint64, C.c_int, C.c_int]
    lib.orc_prove.argtypes = [C.POINTER(Circuit), vp, vp, vp, vp, vp, C.c_int, vp, C.POINTER(C.c_uint64), vp]
    lib.orc_msm_fast.argtypes = [C.c_int, vp, vp, C.c_uint64, C.c_int, vp]
    lib.orc_fast_setup.argtypes = [C.POINTER(Circuit), C.c_int, C.POINTER(vp)]
    lib.orc_fast_prove.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int, vp, C.POINTER(C.c_uint64), vp]
    lib.orc_fast_free.argtypes = [vp]; lib.orc_fast_free.restype = None
    lib.orc_fast_setup_ex.argtypes = [C.POINTER(CircuitEx), C.c_int, C.POINTER(vp)]
    lib.orc_fast_prove_ex.argtypes = [vp, vp, vp, vp, vp, vp, vp, C.c_int, vp, C.POINTER(C.c_uint64), vp]
    return lib


def prove(lib, curve_id: int, n: int, nb_public: int, srs: bytes, cols, perm, L: bytes, R: bytes, O: bytes, pub: bytes,
          blinding: bytes, threads: int = 1):
    """cols = (ql, qr, qm, qo, qk) Montgomery byte strings; perm = list of ints (3n).  Returns (rc, blob, challenges)."""
    keep = [srs] + list(cols)
    permarr = (C.c_int64 * len(perm))(*perm)
    c = Circuit()
    c.curve, c.n, c.nb_public = curve_id, n, nb_public
    c.srs = C.cast(C.c_char_p(srs), C.c_void_p)
    c.ql, c.qr, c.qm, c.qo, c.qk = (C.cast(C.c_char_p(x), C.c_void_p) for x in cols)
    c.perm = C.cast(permarr, C.c_void_p)
    blob = C.create_string_buffer(1200)
    ln = C.c_uint64(0)
    ch = C.create_string_buffer(160)
    rc = lib.orc_prove(C.byref(c), L, R, O, pub, blinding, threads, blob, C.byref(ln), ch)
    del keep
    return rc, blob.raw[: ln.value], [int.from_bytes(ch.raw[32 * i: 32 * i + 32], "big") for i in range(5)]


class FastProver:
    """The performance-first host prover (oracle/fast_prover.c): circuit-only work once, then proofs with the bytes of `prove`."""

    def __init__(self, lib, curve_id: int, n: int, nb_public: int, srs: bytes, cols, perm, threads: int = 1,
                 qcp=(), cci=(), srs_lagrange: bytes = None):
        """qcp / cci / srs_lagrange: the BSB22 part (Qcp columns in Lagrange form, VK CommitmentConstraintIndexes, Lagrange SRS)."""
        self.lib = lib
        self._keep = [srs, srs_lagrange] + list(cols) + list(qcp)
        permarr = (C.c_int64 * len(perm))(*perm)
        e = CircuitEx()
        c = e.base
        c.curve, c.n, c.nb_public = curve_id, n, nb_public
        c.srs = C.cast(C.c_char_p(srs), C.c_void_p)
        c.ql, c.qr, c.qm, c.qo, c.qk = (C.cast(C.c_char_p(x), C.c_void_p) for x in cols)
        c.perm = C.cast(permarr, C.c_void_p)
        e.nb_commit = len(qcp)
        for k, col in enumerate(qcp):
            e.qcp[k] = C.cast(C.c_char_p(col), C.c_void_p)
            e.cci[k] = cci[k]
        if srs_lagrange:
            e.srs_lagrange = C.cast(C.c_char_p(srs_lagrange), C.c_void_p)
        self._ctx = C.c_void_p()
        rc = lib.orc_fast_setup_ex(C.byref(e), threads, C.byref(self._ctx))
        if rc != 0:
            raise RuntimeError("orc_fast_setup_ex returned %d" % rc)

    def prove(self, L: bytes, R: bytes, O: bytes, pub: bytes, blinding: bytes, threads: int = 1, pi2=()):
        blob = C.create_string_buffer(1600)
        ln = C.c_uint64(0)
        ch = C.create_string_buffer(160)
        arr = (C.c_void_p * max(len(pi2), 1))(*[C.cast(C.c_char_p(p), C.c_void_p) for p in pi2]) if pi2 else None
        rc = self.lib.orc_fast_prove_ex(self._ctx, L, R, O, pub, blinding, arr, threads, blob, C.byref(ln), ch)
        return rc, blob.raw[: ln.value], [int.from_bytes(ch.raw[32 * i: 32 * i + 32], "big") for i in range(5)]

    def close(self):
        if self._ctx:
            self.lib.orc_fast_free(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
