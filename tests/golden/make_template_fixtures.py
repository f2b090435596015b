#!/usr/bin/env python3
"""Pins the oracle to the REFERENCE'S OWN verifier: renders the reference's AVM verifier templates and executes them.

BUILD CONTAINER ONLY (reads /root/reference); the output, tests/golden/template_verdicts.json, is data: inputs (verifying
key, proof blob, public-input blob), the template's verdict and its intermediate values.  No reference text is stored.

What runs
---------
The reference's logicsig verifiers are Python source held in Go string constants
(/root/reference/verifier/templateLogicSigBN254.go:3-399, templateLogicSigBLS12_381.go:3-422) and rendered with Go's
text/template by verifier.WritePythonCode (/root/reference/verifier/verifier.go:37-122; funcmap `inc/add/mul/contractName/
frstr/fpstr/hex/hexEncoded` at :43-106).  This script
  1. reads the template text out of the Go files (the raw string literal between the back quotes),
  2. renders it with a small text/template interpreter (`GoTemplate` below: actions, `range $i, $e := pipeline`, `if`,
     `end`, `{{-`/`-}}` trimming, parenthesised calls, `.Field` / `$.Field` / `$var.Field`, `len`, `gt`) and the funcmap of
     verifier.go restated as Python callables,
  3. executes the rendered program under a minimal `algopy` shim (`BigUInt`, `Bytes`, `UInt64`, `urange`, `arc4.UInt256`,
     `arc4.DynamicArray`, `op.sha256/bzero/setbit_bytes`, `op.EllipticCurve.add/scalar_mul/pairing_check`, `Txn`, `Global`) that
     follows the AVM's semantics for the opcodes the template reaches (byte-math results carry no leading zeros, `b|` pads on
     the left, 64-byte operand limit, ec ops fail on malformed points, pairing_check also on points outside the subgroup),
     with the curve arithmetic supplied by oracle/curves.py and oracle/pairing_*.py (themselves pinned by the reference's
     trusted-setup known answers, tests/test_oracle_kat.py),
  4. runs `verify()` on oracle proofs of the reference's test circuits for k = 0, 1, 2 BSB22 commitments on both curves, and
     on the reference's mutations (testutils/verifier_integration_test.go:188-228: flipped public-input byte, first G1
     point := second G1 point; :232-256: rekey) plus a non-canonical scalar and a truncated proof,
  5. stores verdicts and intermediates (gamma, beta, alpha, zeta, PI, lin(zeta), [lin], folded digest, claims),
  6. renders and runs the SMART-CONTRACT flavour of each verifier too (templateSmartContractBN254.go / ...BLS12_381.go: an ARC4
     contract whose `verify(proof: DynamicArray[Bytes32], public_inputs: DynamicArray[Bytes32]) -> arc4.Bool` holds the same
     logic over 32-byte words) on the same inputs and insists on the same verdict and intermediates: all four templates of the
     reference are executed, and they agree with each other.
tests/test_template_pin.py then holds oracle/plonk.py::verify, libapk's apk_verify and the HIP prover to this file.
"""
from __future__ import annotations

import hashlib
import json
import os
import re
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"

from oracle import circuits as ocircuits, curves as ocurves, plonk as oplonk   # noqa: E402
from oracle.prng import SplitMix64, tau_from_seed                               # noqa: E402


# ======================================================================================================================
# 1. Go text/template, the subset the four templates use
# ======================================================================================================================

def go_raw_string(path: str) -> str:
    src = open(path, encoding="utf-8").read()
    a = src.index("`")
    b = src.rindex("`")
    return src[a + 1: b]


class GoTemplate:
    """text/template subset.  Nodes: ("text", s) | ("action", expr) | ("range", ivar, evar, expr, body) | ("if", expr, body)."""

    # `{{/* ... */}}` comments may contain `}}` (templateSmartContractBN254.go:133 wraps a py.log call in one): matched first
    ACTION = re.compile(r"\{\{(-\s)?\s*(/\*.*?\*/|.*?)\s*(\s-)?\}\}", re.S)

    def __init__(self, text: str, funcs: dict):
        self.funcs = dict(funcs)
        self.funcs.setdefault("len", len)
        self.funcs.setdefault("gt", lambda a, b: a > b)
        toks = []
        pos = 0
        for m in self.ACTION.finditer(text):
            lit = text[pos: m.start()]
            if m.group(1):
                lit = lit.rstrip(" \t\r\n")
            toks.append(("text", lit))
            toks.append(("act", m.group(2), bool(m.group(3))))
            pos = m.end()
        toks.append(("text", text[pos:]))
        # right-trim markers eat the whitespace that follows
        for i, t in enumerate(toks):
            if t[0] == "act" and t[2]:
                nxt = toks[i + 1]
                toks[i + 1] = ("text", nxt[1].lstrip(" \t\r\n"))
        self.toks = toks
        self.i = 0
        self.tree = self._parse_block(top=True)
        assert self.i == len(self.toks), "unbalanced {{ end }}"

    def _parse_block(self, top=False):
        out = []
        while self.i < len(self.toks):
            t = self.toks[self.i]
            self.i += 1
            if t[0] == "text":
                if t[1]:
                    out.append(("text", t[1]))
                continue
            body = t[1]
            if body.startswith("/*"):
                continue
            if body == "end":
                assert not top, "{{ end }} without a range / if"
                return out
            if body.startswith("range"):
                m = re.match(r"range\s+(\$\w+)\s*,\s*(\$\w+)\s*:=\s*(.*)$", body, re.S)
                assert m, body
                out.append(("range", m.group(1), m.group(2), m.group(3), self._parse_block()))
            elif body.startswith("if"):
                out.append(("if", body[2:].strip(), self._parse_block()))
            else:
                assert not body.startswith(("else", "with", "define", "template", "block")), "unsupported action: " + body
                out.append(("action", body))
        assert top, "range / if without its {{ end }}"
        return out

    # ---- expressions -------------------------------------------------------------------------------------------------
    TOKEN = re.compile(r"\s*(\(|\)|\$?[\w.]*[\w]|\$|\.)")

    def _tokens(self, s: str):
        out, pos = [], 0
        s = s.strip()
        while pos < len(s):
            m = self.TOKEN.match(s, pos)
            assert m and m.end() > pos, "cannot tokenise %r at %d" % (s, pos)
            out.append(m.group(1))
            pos = m.end()
        return out

    def _eval(self, s: str, dot, env):
        toks = self._tokens(s)
        val, rest = self._command(toks, dot, env)
        assert not rest, (s, rest)
        return val

    def _command(self, toks, dot, env):
        terms = []
        while toks and toks[0] != ")":
            if toks[0] == "(":
                v, toks = self._command(toks[1:], dot, env)
                assert toks and toks[0] == ")"
                toks = toks[1:]
                terms.append(("val", v))
            else:
                terms.append(("tok", toks[0]))
                toks = toks[1:]
        assert terms
        head = terms[0]
        if head[0] == "tok" and head[1] in self.funcs:
            args = [self._term(t, dot, env) for t in terms[1:]]
            return self.funcs[head[1]](*args), toks
        assert len(terms) == 1, terms
        return self._term(head, dot, env), toks

    def _term(self, t, dot, env):
        if t[0] == "val":
            return t[1]
        tok = t[1]
        if re.fullmatch(r"-?\d+", tok):
            return int(tok)
        if tok.startswith("$"):
            name, _, path = tok.partition(".")
            base = env["$"] if name == "$" else env[name]
        else:
            assert tok.startswith("."), tok
            base, path = dot, tok[1:]
        for f in [p for p in path.split(".") if p]:
            base = getattr(base, f)
        return base

    # ---- execution ---------------------------------------------------------------------------------------------------
    def render(self, data) -> str:
        out = []
        self._exec(self.tree, data, {"$": data}, out)
        return "".join(out)

    def _exec(self, nodes, dot, env, out):
        for nd in nodes:
            if nd[0] == "text":
                out.append(nd[1])
            elif nd[0] == "action":
                v = self._eval(nd[1], dot, env)
                out.append(str(v))
            elif nd[0] == "if":
                if self._eval(nd[1], dot, env):
                    self._exec(nd[2], dot, env, out)
            else:
                _, ivar, evar, expr, body = nd
                for i, e in enumerate(self._eval(expr, dot, env)):
                    sub = dict(env)
                    sub[ivar], sub[evar] = i, e
                    self._exec(body, e, sub, out)       # text/template sets dot to the element inside range


class NS:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def vk_view(ov, vk: oplonk.VerifyingKey, g2):
    """The fields of gnark's plonk VerifyingKey that the templates dereference (App. A.6 of SURVEY.md)."""
    def g1(Pt):
        return NS(pt=Pt, X=0 if Pt is None else Pt[0], Y=0 if Pt is None else Pt[1])

    def e2(c):
        return NS(A0=c[0], A1=c[1])

    return NS(Kzg=NS(G1=g1(vk.g1), G2=[NS(X=e2(Q[0]), Y=e2(Q[1])) for Q in g2]),
              NbPublicVariables=vk.nb_public, Size=vk.size, SizeInv=vk.size_inv, Generator=vk.generator,
              CosetShift=vk.coset_shift, Ql=g1(vk.ql), Qr=g1(vk.qr), Qm=g1(vk.qm), Qo=g1(vk.qo), Qk=g1(vk.qk),
              S=[g1(s) for s in vk.s], Qcp=[g1(q) for q in vk.qcp],
              CommitmentConstraintIndexes=list(vk.commitment_constraint_indexes))


def funcmap(ov):
    """verifier/verifier.go:43-106.  `hex` = p.RawBytes(); the BLS12-381 flavour rewrites the infinity flag 0x40 to 0x00
    (:92-101) and `hexEncoded` keeps it (:102-105).  The BN254 funcmap has no hexEncoded and no rewrite: the same bytes go
    to the hash and to the AVM's ec ops, so infinity must be the all-zero encoding there."""
    n = ov.fp_bytes

    def raw(g):
        if g.pt is None:
            return bytes([0x40 if ov.name == "bls12-381" else 0x00]) + bytes(2 * n - 1)
        return g.X.to_bytes(n, "big") + g.Y.to_bytes(n, "big")

    def hex_(g):
        b = bytearray(raw(g))
        if g.pt is None:
            b[0] = 0
        return bytes(b).hex()

    fm = {"inc": lambda i: i + 1, "add": lambda a, b: int(a) + int(b), "mul": lambda a, b: int(a) * int(b),
          "contractName": lambda: "Verifier", "frstr": lambda x: str(int(x)), "fpstr": lambda x: str(int(x)), "hex": hex_}
    if ov.name == "bls12-381":
        fm["hexEncoded"] = lambda g: raw(g).hex()
    return fm


# ======================================================================================================================
# 2. algopy shim (AVM semantics for what the templates reach)
# ======================================================================================================================

class AvmError(Exception):
    """The AVM would fail the program ('rejected by logic')."""


def _i(x) -> int:
    if isinstance(x, (BigUInt, UInt64)):
        return x.v
    if isinstance(x, bool):
        return int(x)
    if isinstance(x, int):
        return x
    raise TypeError(type(x))


class UInt64:
    def __init__(self, v=0):
        v = _i(v)
        if not 0 <= v < 1 << 64:
            raise AvmError("uint64 overflow")
        self.v = v

    def _b(self, o, f):
        return UInt64(f(self.v, _i(o)))

    __add__ = lambda s, o: s._b(o, lambda a, b: a + b)
    __radd__ = __add__
    __mul__ = lambda s, o: s._b(o, lambda a, b: a * b)
    __rmul__ = __mul__
    __floordiv__ = lambda s, o: s._b(o, lambda a, b: a // b)
    __mod__ = lambda s, o: s._b(o, lambda a, b: a % b)

    def __sub__(self, o):
        if self.v < _i(o):
            raise AvmError("uint64 underflow")
        return UInt64(self.v - _i(o))

    __eq__ = lambda s, o: s.v == _i(o)
    __ne__ = lambda s, o: s.v != _i(o)
    __lt__ = lambda s, o: s.v < _i(o)
    __le__ = lambda s, o: s.v <= _i(o)
    __gt__ = lambda s, o: s.v > _i(o)
    __ge__ = lambda s, o: s.v >= _i(o)
    __hash__ = lambda s: hash(s.v)
    __index__ = lambda s: s.v
    __bool__ = lambda s: s.v != 0
    __int__ = lambda s: s.v


def urange(*a):
    for v in range(*[_i(x) for x in a]):
        yield UInt64(v)


class Bytes:
    def __init__(self, b=b""):
        b = b.b if isinstance(b, Bytes) else bytes(b)
        if len(b) > 4096:
            raise AvmError("byte string longer than 4096")
        self.b = b

    @staticmethod
    def from_hex(h: str) -> "Bytes":
        return Bytes(bytes.fromhex(h))

    @property
    def length(self) -> UInt64:
        return UInt64(len(self.b))

    def __add__(self, o):
        return Bytes(self.b + _bytes(o))

    def __radd__(self, o):
        return Bytes(_bytes(o) + self.b)

    def __getitem__(self, k):
        if isinstance(k, slice):
            assert k.step is None
            n = len(self.b)

            def clamp(v, default):
                if v is None:
                    return default
                v = _i(v)
                if v < 0:
                    v += n
                return min(max(v, 0), n)

            lo, hi = clamp(k.start, 0), clamp(k.stop, n)
            return Bytes(self.b[lo: max(lo, hi)])
        k = _i(k)
        if not -len(self.b) <= k < len(self.b):
            raise AvmError("index out of range")
        return Bytes(self.b[k: k + 1] if k >= 0 else self.b[k:][:1])

    def _pad(self, o):
        a, b = self.b, _bytes(o)
        n = max(len(a), len(b))
        return a.rjust(n, b"\0"), b.rjust(n, b"\0")

    def __or__(self, o):       # b|  : the shorter operand is zero-extended on the left
        a, b = self._pad(o)
        return Bytes(bytes(x | y for x, y in zip(a, b)))

    __ror__ = __or__

    def __xor__(self, o):      # b^
        a, b = self._pad(o)
        return Bytes(bytes(x ^ y for x, y in zip(a, b)))

    def __eq__(self, o):
        return self.b == _bytes(o)

    def __ne__(self, o):
        return self.b != _bytes(o)

    __hash__ = lambda s: hash(s.b)
    __bool__ = lambda s: len(s.b) > 0

    @property
    def bytes(self):
        return self


def _bytes(x) -> bytes:
    if isinstance(x, Bytes):
        return x.b
    if isinstance(x, (bytes, bytearray)):
        return bytes(x)
    if isinstance(x, BigUInt):
        return x.bytes.b
    raise TypeError(type(x))


class BigUInt:
    """AVM byte-math values: operands at most 64 bytes, results without leading zero bytes (zero = empty string)."""

    def __init__(self, v=0):
        v = _i(v)
        if v < 0:
            raise AvmError("negative biguint")
        self.v = v

    @staticmethod
    def from_bytes(b) -> "BigUInt":
        return BigUInt(int.from_bytes(_bytes(b), "big"))

    @property
    def bytes(self) -> Bytes:
        return Bytes(self.v.to_bytes((self.v.bit_length() + 7) // 8, "big"))

    @staticmethod
    def _arg(o) -> int:
        v = _i(o)
        if v.bit_length() > 512:
            raise AvmError("math attempted on large byte-array")
        return v

    def _b(self, o, f):
        return BigUInt(f(self._arg(self), self._arg(o)))

    __add__ = lambda s, o: s._b(o, lambda a, b: a + b)
    __radd__ = __add__
    __mul__ = lambda s, o: s._b(o, lambda a, b: a * b)
    __rmul__ = __mul__

    def __sub__(self, o):
        a, b = self._arg(self), self._arg(o)
        if a < b:
            raise AvmError("byte math would have negative result")
        return BigUInt(a - b)

    def __rsub__(self, o):
        return BigUInt(o).__sub__(self)

    def _div(self, o, f):
        if _i(o) == 0:
            raise AvmError("division by zero")
        return self._b(o, f)

    __floordiv__ = lambda s, o: s._div(o, lambda a, b: a // b)
    __mod__ = lambda s, o: s._div(o, lambda a, b: a % b)
    __eq__ = lambda s, o: s.v == _i(o)
    __ne__ = lambda s, o: s.v != _i(o)
    __lt__ = lambda s, o: s.v < _i(o)
    __le__ = lambda s, o: s.v <= _i(o)
    __gt__ = lambda s, o: s.v > _i(o)
    __ge__ = lambda s, o: s.v >= _i(o)
    __hash__ = lambda s: hash(s.v)
    __bool__ = lambda s: s.v != 0
    __int__ = lambda s: s.v


class UInt256:
    """arc4.UInt256: 32-byte big-endian, encoding fails on overflow."""

    def __init__(self, v=0):
        v = _i(v)
        if v >= 1 << 256:
            raise AvmError("uint256 overflow")
        self.v = v

    @property
    def bytes(self) -> Bytes:
        return Bytes(self.v.to_bytes(32, "big"))


class _DynamicArray:
    def __init__(self, *items):
        self.items = list(items)

    def append(self, x):
        self.items.append(x)

    def __iter__(self):
        return iter(list(self.items))

    def _k(self, k):
        k = _i(k)
        if not 0 <= k < len(self.items):
            raise AvmError("array index out of bounds")
        return k

    def __getitem__(self, k):
        return self.items[self._k(k)]

    def __setitem__(self, k, v):
        self.items[self._k(k)] = v

    def __class_getitem__(cls, item):
        return cls


class _Bytes32:
    """arc4.StaticArray[arc4.Byte, Literal[32]]: one 32-byte word of the proof / public-input arrays of the smart-contract
    verifiers (templateSmartContractBN254.go:12)."""

    def __init__(self, b=bytes(32)):
        b = _bytes(b)
        if len(b) != 32:
            raise AvmError("Bytes32 of %d bytes" % len(b))
        self.b = b

    @property
    def bytes(self) -> Bytes:
        return Bytes(self.b)

    def copy(self) -> "_Bytes32":
        return _Bytes32(self.b)


class _StaticArray:
    def __class_getitem__(cls, item):
        return _Bytes32


class _WordArray(_DynamicArray):
    @property
    def length(self) -> UInt64:
        return UInt64(len(self.items))


class _ArcBool:
    def __init__(self, v=False):
        self.native = bool(v)

    def __bool__(self):
        return self.native


def words(blob: bytes) -> _WordArray:
    """utils.ProofAndPublicInputsForAtomicComposer (utils/utils.go:162-172,215-224): the blob as 32-byte words."""
    if len(blob) % 32:
        raise AvmError("blob is not a whole number of 32-byte words")
    return _WordArray(*[_Bytes32(blob[i: i + 32]) for i in range(0, len(blob), 32)])


class Avm:
    """Transaction context + the elliptic-curve opcodes, for one curve."""

    def __init__(self, ov, pairing):
        self.ov, self.pairing = ov, pairing
        self.args = []
        self.rekey_to = bytes(32)

    # -- point decoding as the AVM does it (gnark-crypto behind `ec_*`): X || Y fixed width, all zero = infinity,
    #    coordinates below p, on the curve; the subgroup is only checked by pairing_check.
    def g1(self, b: Bytes, subgroup=False):
        raw = _bytes(b)
        n = self.ov.fp_bytes
        if len(raw) != 2 * n:
            raise AvmError("bad G1 point length %d" % len(raw))
        if not any(raw):
            return None
        x, y = int.from_bytes(raw[:n], "big"), int.from_bytes(raw[n:], "big")
        if x >= self.ov.p or y >= self.ov.p:
            raise AvmError("G1 coordinate not reduced")
        if not self.ov.is_on_curve((x, y)):
            raise AvmError("G1 point not on curve")
        if subgroup and self.ov.mul_raw((x, y), self.ov.r) is not None:
            raise AvmError("G1 point outside the subgroup")
        return (x, y)

    def g1_out(self, Pt) -> Bytes:
        n = self.ov.fp_bytes
        if Pt is None:
            return Bytes(bytes(2 * n))
        return Bytes(Pt[0].to_bytes(n, "big") + Pt[1].to_bytes(n, "big"))

    def g2(self, raw: bytes):
        n = self.ov.fp_bytes
        c = [int.from_bytes(raw[i * n:(i + 1) * n], "big") for i in range(4)]
        if any(v >= self.ov.p for v in c):
            raise AvmError("G2 coordinate not reduced")
        if not any(c):
            return None
        Q = ((c[0], c[1]), (c[2], c[3]))          # X.A0 || X.A1 || Y.A0 || Y.A1
        if not self.pairing.g2_on_curve(Q):
            raise AvmError("G2 point not on the twist")
        if self.pairing.g2_mul_raw(Q, self.ov.r) is not None:
            raise AvmError("G2 point outside the subgroup")
        return Q


def install_algopy(avm: Avm):
    """Builds `algopy`, `algopy.arc4`, `algopy.op` modules around one Avm and registers them in sys.modules."""
    ov = avm.ov
    curve_tag = "BN254g1" if ov.name == "bn254" else "BLS12_381g1"

    class EC:
        BN254g1 = "BN254g1"
        BLS12_381g1 = "BLS12_381g1"

    class EllipticCurve:
        @staticmethod
        def _g(g):
            if g != curve_tag:
                raise AvmError("wrong curve group " + str(g))

        @staticmethod
        def add(g, a, b):
            EllipticCurve._g(g)
            return avm.g1_out(ov.add(avm.g1(a), avm.g1(b)))

        @staticmethod
        def scalar_mul(g, a, s):
            EllipticCurve._g(g)
            sb = _bytes(s)
            if len(sb) > 32:
                raise AvmError("ec_scalar_mul scalar longer than 32 bytes")
            return avm.g1_out(ov.mul_raw(avm.g1(a), int.from_bytes(sb, "big")))

        @staticmethod
        def pairing_check(g, a, b):
            EllipticCurve._g(g)
            a, b = _bytes(a), _bytes(b)
            n = ov.fp_bytes
            if len(a) % (2 * n) or len(b) % (4 * n) or len(a) // (2 * n) != len(b) // (4 * n):
                raise AvmError("pairing_check operand sizes")
            k = len(a) // (2 * n)
            ps = [avm.g1(Bytes(a[i * 2 * n:(i + 1) * 2 * n]), subgroup=True) for i in range(k)]
            qs = [avm.g2(b[i * 4 * n:(i + 1) * 4 * n]) for i in range(k)]
            pairs = [(p, q) for p, q in zip(ps, qs) if p is not None and q is not None]
            if not pairs:
                return True
            return avm.pairing.pairing_check([p for p, _ in pairs], [q for _, q in pairs])

    def sha256(b):
        return Bytes(hashlib.sha256(_bytes(b)).digest())

    def bzero(n):
        return Bytes(bytes(_i(n)))

    def setbit_bytes(b, idx, val):
        raw = bytearray(_bytes(b))
        idx = _i(idx)
        if idx >= 8 * len(raw):
            raise AvmError("setbit index out of range")
        mask = 0x80 >> (idx % 8)                     # bit 0 = most significant bit of byte 0
        raw[idx // 8] = (raw[idx // 8] | mask) if val else (raw[idx // 8] & ~mask)
        return Bytes(bytes(raw))

    class _TxnMeta(type):
        rekey_to = property(lambda cls: Bytes(avm.rekey_to))

    Txn = _TxnMeta("Txn", (), {"application_args": staticmethod(lambda i: Bytes(avm.args[_i(i)]))})

    class Global:
        zero_address = Bytes(bytes(32))
        creator_address = Bytes(bytes([1]) * 32)

    def logicsig(name=None):
        return lambda f: f

    def abimethod(*a, **kw):                    # @abimethod and @abimethod(create='require', ...)
        if len(a) == 1 and callable(a[0]) and not kw:
            return a[0]
        return lambda f: f

    class ARC4Contract:
        pass

    algopy = types.ModuleType("algopy")
    arc4 = types.ModuleType("algopy.arc4")
    op = types.ModuleType("algopy.op")
    arc4.UInt256, arc4.DynamicArray = UInt256, _DynamicArray
    arc4.abimethod, arc4.StaticArray, arc4.String, arc4.Byte, arc4.Bool = abimethod, _StaticArray, str, int, _ArcBool
    op.bzero, op.sha256, op.EllipticCurve, op.EC, op.setbit_bytes = bzero, sha256, EllipticCurve, EC, setbit_bytes
    for k, v in dict(logicsig=logicsig, subroutine=lambda f: f, BigUInt=BigUInt, Bytes=Bytes, UInt64=UInt64, urange=urange,
                     arc4=arc4, op=op, Txn=Txn, Global=Global, ARC4Contract=ARC4Contract, log=lambda *a: None).items():
        setattr(algopy, k, v)
    sys.modules["algopy"], sys.modules["algopy.arc4"], sys.modules["algopy.op"] = algopy, arc4, op


def run_template(program: str, avm: Avm, proof: bytes, public_inputs: bytes, rekey: bool = False, contract: bool = False):
    """One simulated logicsig evaluation.  app args = [method selector, arc4 byte[32][] proof, arc4 byte[32][] public inputs]
    (utils/utils.go:162-172,196-224; the logicsig strips the 2-byte count, templateLogicSigBN254.go:46-47)."""
    install_algopy(avm)
    arc4_words = lambda b: (len(b) // 32).to_bytes(2, "big") + b
    avm.args = [b"\0\0\0\0", arc4_words(proof), arc4_words(public_inputs)]
    avm.rekey_to = bytes([7]) * 32 if rekey else bytes(32)
    g = {"__name__": "rendered_verifier"}
    exec(compile(program, "<rendered template>", "exec"), g)
    captured = {}

    def tracer(frame, event, arg):
        if frame.f_code.co_name != "verify":
            return None

        def local(fr, ev, a):
            loc = fr.f_locals
            if ev in ("return", "exception"):
                captured.update(loc)
            elif ev == "line":
                # `r` is reused: its first value is the folding challenge (templateLogicSigBN254.go:287); `digest` / `claims`
                # are the folded opening until the second sha256 (:323) batches the two openings with verifier-side randomness
                if "r_acc" in loc and "gamma_kzg" not in captured:
                    captured["gamma_kzg"] = loc["r"]
                if "quotient" not in loc and "digest" in loc and "claims" in loc:
                    captured["folded_digest"], captured["folded_claims"] = loc["digest"], loc["claims"]
            return local
        return local

    sys.settrace(tracer)
    try:
        if contract:
            # the ARC4 method `verify(proof: DynamicArray[Bytes32], public_inputs: DynamicArray[Bytes32]) -> arc4.Bool` of the
            # smart-contract flavour (templateSmartContractBN254.go:54-58); a False return is what the caller's assert turns into a
            # failed transaction (testutils/verifier_integration_test.go:379-437)
            ok = bool(g["Verifier"]().verify(words(proof), words(public_inputs)))
        else:
            ok = g["verify"]()
        verdict, why = ("accept", "") if ok else ("reject", "returned False")
    except AssertionError:
        verdict, why = "reject", "assert"
    except AvmError as e:
        verdict, why = "reject", "avm: " + str(e)
    finally:
        sys.settrace(None)
    return verdict, why, captured


def intermediates(loc: dict) -> dict:
    out = {}
    for k in ("gamma", "beta", "alpha", "zeta", "PI", "linearized_poly_at_z", "gamma_kzg", "folded_claims", "claims"):
        if k in loc:
            out[k] = hex(_i(loc[k]))
    for k in ("lin_poly_com", "folded_h", "folded_digest", "digest", "quotient"):
        if k in loc:
            out[k] = _bytes(loc[k]).hex()
    return out


# ======================================================================================================================
# 3. cases
# ======================================================================================================================

def pairing_module(ov):
    if ov.name == "bn254":
        from oracle import pairing_bn254 as pr
    else:
        from oracle import pairing_bls12381 as pr
    if not hasattr(pr, "g2_mul_raw"):
        def g2_mul_raw(Q, k):
            acc = None
            for bit in bin(k)[2:]:
                acc = pr.g2_add(acc, acc)
                if bit == "1":
                    acc = pr.g2_add(acc, Q)
            return acc
        pr.g2_mul_raw = g2_mul_raw
    return pr


def ensure_mul_raw(ov):
    """[k]P without reducing k mod r (the AVM multiplies by the integer; needed for the subgroup test [r]P)."""
    if hasattr(ov, "mul_raw"):
        return

    def mul_raw(Pt, k):
        acc = None
        for bit in bin(k)[2:] if k else "":
            acc = ov.add(acc, acc)
            if bit == "1":
                acc = ov.add(acc, Pt)
        return acc
    object.__setattr__(ov, "mul_raw", mul_raw)


CASES = [
    # name, builder, tau seed, blinding seed
    ("pythagorean", lambda ov: ocircuits.pythagorean(ov) + (None,), 0x7E57, 5),
    ("identity", lambda ov: ocircuits.identity(ov) + (None,), 0x7E58, 6),
    ("random_chain_2p3", lambda ov: ocircuits.random_chain(ov, 3, 0xA190) + (None,), 1, 7),
    ("bsb22_square_k1", lambda ov: ocircuits.bsb22_square(ov, 1), 0xB5B, 8),
    ("bsb22_square_k2", lambda ov: ocircuits.bsb22_square(ov, 2), 0xB5C, 9),
]


def pt_json(ov, Pt):
    return None if Pt is None else [hex(Pt[0]), hex(Pt[1])]


def build_case(ov, name, builder, tau_seed, bl_seed):
    r = ov.r
    c, sol, plan = builder(ov)
    n = c.domain_size()
    tau = tau_from_seed(tau_seed, r)
    osrs = oplonk.synthetic_srs(ov, n, tau, materialize=False)
    opk = oplonk.setup(c, osrs)
    pi2 = None
    hiding = []
    if plan is not None:
        hiding = [(11 + i, 22 + i) for i in range(len(plan))]
        wn = ov.omega(n)
        sol, pi2 = ocircuits.solve_bsb22(c, sol, plan, lambda col: osrs.commit(oplonk.intt(col, wn, r)), hiding)
    L, R, O = oplonk.solve_lro(c, sol)
    pub = sol[: c.nb_public]
    g = SplitMix64(bl_seed)
    bl = [g.fr(r) for _ in range(9)]
    pr = oplonk.prove(opk, L, R, O, pub, bl, pi2=pi2) if pi2 is not None else oplonk.prove(opk, L, R, O, pub, bl)
    blob = oplonk.marshal_proof(ov, pr)
    pib = oplonk.marshal_public_inputs(pub)
    return dict(name=name, tau=tau, tau_seed=tau_seed, blinding_seed=bl_seed, hiding=hiding, vk=opk.vk, proof=blob, public=pib,
                nb_constraints=len(c.constraints))


def mutations(ov, blob: bytes, pib: bytes):
    """(label, proof, public inputs, rekey)"""
    w = 2 * ov.fp_bytes
    out = [("valid", blob, pib, False)]
    if pib:
        p2 = bytearray(pib)
        p2[0] = 1 if p2[0] == 0 else 0                                   # verifier_integration_test.go:190-195
        out.append(("public_input_byte_flipped", blob, bytes(p2), False))
    b2 = bytearray(blob)
    b2[0:w] = blob[w: 2 * w]                                             # :217-219
    out.append(("first_g1_overwritten_by_second", bytes(b2), pib, False))
    out.append(("rekey", blob, pib, True))                               # :232-256
    off = 6 * w                                                          # l(zeta): add r so the scalar is not canonical
    v = int.from_bytes(blob[off: off + 32], "big") + ov.r
    if v < 1 << 256:
        b3 = bytearray(blob)
        b3[off: off + 32] = v.to_bytes(32, "big")
        out.append(("claimed_value_plus_r", bytes(b3), pib, False))
    out.append(("proof_truncated_by_one_word", blob[:-32], pib, False))
    b4 = bytearray(blob)
    b4[off + 31] ^= 1                                                    # l(zeta) changed, still canonical
    out.append(("claimed_value_bit_flipped", bytes(b4), pib, False))
    b5 = bytearray(blob)
    zoff = 6 * w + 5 * 32                                                # [Z]: last byte of Y changed -> not a curve point
    b5[zoff + w - 1] ^= 1
    out.append(("z_commitment_off_curve", bytes(b5), pib, False))
    return out


def old_infinity_encoding_control():
    """What the pin found, kept as a fixture: rounds 1-2 hashed infinity as 0x40 00.. on BOTH curves.  A BN254 proof made that
    way (the prover's transcript sees 0x40.. for [Qk] = infinity of the Pythagorean circuit) is REJECTED by the executed BN254
    template, which hashes the all-zero constant it also feeds to the ec ops."""
    ov = ocurves.BN254
    ensure_mul_raw(ov)
    pr = pairing_module(ov)
    tmpl = GoTemplate(go_raw_string(os.path.join(REF, "verifier", "templateLogicSigBN254.go")), funcmap(ov))
    good_raw = type(ov).raw_bytes

    def raw_bytes_0x40(self, P):
        if P is None:
            return bytes([0x40]) + bytes(2 * self.fp_bytes - 1)
        return good_raw(self, P)

    type(ov).raw_bytes = raw_bytes_0x40
    try:
        cs = build_case(ov, *CASES[0])                  # pythagorean: [Qk] is the point at infinity
    finally:
        type(ov).raw_bytes = good_raw
    g2 = [pr.G2_GEN, pr.g2_mul(pr.G2_GEN, cs["tau"])]
    verdict, why, loc = run_template(tmpl.render(vk_view(ov, cs["vk"], g2)), Avm(ov, pr), cs["proof"], cs["public"])
    assert verdict == "reject", "the 0x40 encoding of infinity was expected to fail the BN254 template"
    T = {}
    assert not oplonk.verify(cs["vk"], cs["proof"], cs["public"], T)
    print("negative control: BN254 proof with 0x40-encoded infinity in the transcript ->", verdict, why)
    return {"what": "BN254 pythagorean proof whose prover hashed infinity as 0x40 00.. (the encoding of rounds 1-2)", "proof": cs["proof"].hex(),
            "public_inputs": cs["public"].hex(), "verdict": verdict, "why": why, "gamma_of_the_template": intermediates(loc).get("gamma")}


def main():
    out = {"generator": "tests/golden/make_template_fixtures.py",
           "what": "verdicts and intermediates of the reference's rendered logicsig verifier templates, executed under an algopy/AVM shim",
           "templates": {}, "cases": []}
    for ov, fname, sc_name in ((ocurves.BN254, "templateLogicSigBN254.go", "templateSmartContractBN254.go"),
                               (ocurves.BLS12_381, "templateLogicSigBLS12_381.go", "templateSmartContractBLS12_381.go")):
        ensure_mul_raw(ov)
        pr = pairing_module(ov)
        path, sc_path = os.path.join(REF, "verifier", fname), os.path.join(REF, "verifier", sc_name)
        vgo = hashlib.sha256(open(os.path.join(REF, "verifier", "verifier.go"), "rb").read()).hexdigest()
        for nm, pth in ((fname, path), (sc_name, sc_path)):
            out["templates"][nm] = {"sha256": hashlib.sha256(open(pth, "rb").read()).hexdigest(), "verifier.go_sha256": vgo}
        tmpl = GoTemplate(go_raw_string(path), funcmap(ov))
        sc_tmpl = GoTemplate(go_raw_string(sc_path), funcmap(ov))
        for name, builder, tau_seed, bl_seed in CASES:
            cs = build_case(ov, name, builder, tau_seed, bl_seed)
            vk = cs["vk"]
            g2 = [pr.G2_GEN, pr.g2_mul(pr.G2_GEN, cs["tau"])]
            program = tmpl.render(vk_view(ov, vk, g2))
            sc_program = sc_tmpl.render(vk_view(ov, vk, g2))
            results = []
            for label, blob, pib, rekey in mutations(ov, cs["proof"], cs["public"]):
                verdict, why, loc = run_template(program, Avm(ov, pr), blob, pib, rekey)
                print("%-10s %-18s %-32s %s %s" % (ov.name, name, label, verdict, why), flush=True)
                results.append({"mutation": label, "proof": blob.hex(), "public_inputs": pib.hex(), "rekey": rekey,
                                "verdict": verdict, "why": why, "intermediates": intermediates(loc)})
                if not rekey:
                    # the smart-contract flavour of the same verifier (the rekey guard is a logicsig matter): same verdict, same
                    # intermediates, or the two reference templates disagree with each other
                    sc_verdict, sc_why, sc_loc = run_template(sc_program, Avm(ov, pr), blob, pib, False, contract=True)
                    assert sc_verdict == verdict, (ov.name, name, label, verdict, why, sc_verdict, sc_why)
                    a, b = intermediates(loc), intermediates(sc_loc)
                    assert all(a[k] == b[k] for k in a if k in b), (ov.name, name, label)
                    results[-1]["smart_contract"] = {"verdict": sc_verdict, "why": sc_why}
            assert results[0]["verdict"] == "accept" or name == "identity", (ov.name, name, results[0])
            out["cases"].append({
                "curve": ov.name, "circuit": name, "tau_seed": tau_seed, "blinding_seed": bl_seed,
                "bsb22_hiding": [list(h) for h in cs["hiding"]],
                "rendered_program_sha256": hashlib.sha256(program.encode()).hexdigest(),
                "rendered_smart_contract_sha256": hashlib.sha256(sc_program.encode()).hexdigest(),
                "vk": {"size": vk.size, "size_inv": hex(vk.size_inv), "generator": hex(vk.generator), "coset_shift": vk.coset_shift,
                       "nb_public": vk.nb_public, "ql": pt_json(ov, vk.ql), "qr": pt_json(ov, vk.qr), "qm": pt_json(ov, vk.qm),
                       "qo": pt_json(ov, vk.qo), "qk": pt_json(ov, vk.qk), "s": [pt_json(ov, s) for s in vk.s],
                       "qcp": [pt_json(ov, q) for q in vk.qcp], "commitment_constraint_indexes": list(vk.commitment_constraint_indexes),
                       "g1": pt_json(ov, vk.g1),
                       "g2": [[[hex(c) for c in Q[0]], [hex(c) for c in Q[1]]] for Q in g2]},
                "results": results})
    out["negative_control"] = old_infinity_encoding_control()
    json.dump(out, open(os.path.join(HERE, "template_verdicts.json"), "w"), indent=1)
    print("wrote template_verdicts.json:", len(out["cases"]), "cases")


if __name__ == "__main__":
    main()
