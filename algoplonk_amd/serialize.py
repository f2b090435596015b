"""Serialized-artifact interop: the byte formats behind /root/reference/utils/utils.go:88-157.

`utils.SerializeCompiledCircuit` gob-encodes `CompiledCircuitBytes{Ccs, Pk, Vk []byte; Curve ecc.ID}` where the three byte
slices are gnark's `WriteTo` outputs; `DeserializeCompiledCircuit` reverses it.  This module mirrors those two functions
(same names) and the layers underneath.  WHAT IS INTEROPERABLE WITH A GO PROCESS TODAY: the gob envelope and the kzg SRS
encodings (pinned by the reference's own files).  What is NOT: the plonk Vk / Pk blobs (restated from memory, unpinned: the reader
accepts exactly what this writer emits and refuses everything else - round 6 removed the tolerance for a guessed `Kzg.Lines`
block: no layout this package cannot pin is read) and the
constraint system (own tagged encoding; gnark's file is CBOR + intcomp-compressed blocks).  Files written
here are therefore read back HERE; a circuit compiled by Go reaches libapk through the cgo shim, not a file.

layer                                    | status
-----------------------------------------+----------------------------------------------------------------------------------
gob envelope (Go encoding/gob wire format)| written from the gob specification; checked against the spec's own worked example
                                          | (tests/test_serialize.py), byte-exact for this one struct type
kzg.ProvingKey / kzg.VerifyingKey WriteTo | PINNED by the reference: setup/<name>/pk.bin and vk.bin ARE these encodings
                                          | (setup/setup.go:173-174,189-190,196-228; KATs setup/trusted_setup_test.go)
plonk.VerifyingKey / ProvingKey WriteTo   | UNPINNED: field order restated from gnark v0.15.0 backend/plonk/<curve>/marshal.go
                                          | [UPSTREAM, not vendored]; tools/gnark_dump pins it: its JSON carries gnark's own
                                          | vk.WriteTo bytes and ecc.ID value, compared in tests/test_gnark_dump.py
constraint system (ccs.WriteTo)           | gnark's CBOR of SparseR1CS: NOT parsed (opaque bytes are carried through); a circuit
                                          | this package compiled is stored in its own tagged encoding instead

Encodings (gnark-crypto `Encoder`): integers big-endian; Fr 32 bytes big-endian canonical; G1 / G2
compressed with the flag bits in the top of byte 0; slices carry a big-endian u32 length.
"""
from __future__ import annotations

import io
import json
import struct
from dataclasses import dataclass
from typing import List, Optional, Tuple

from . import ecc, frontend, plonk, setup

# ecc.ID values of gnark-crypto v0.20.1 ecc/ecc.go [UPSTREAM]: UNKNOWN = 0, BN254 = 1, BLS12_377 = 2, BLS12_381 = 3 ...
ECC_ID = {"bn254": 1, "bls12_381": 3}
ECC_BY_ID = {1: ecc.BN254, 3: ecc.BLS12_381}


# ---- Go encoding/gob, just enough for CompiledCircuitBytes -------------------------------------------------------------------
def _gob_uint(x: int) -> bytes:
    if x < 128:
        return bytes([x])
    b = x.to_bytes((x.bit_length() + 7) // 8, "big")
    return bytes([256 - len(b)]) + b


def _gob_int(x: int) -> bytes:
    return _gob_uint((~x << 1) | 1 if x < 0 else x << 1)


class _GobReader:
    def __init__(self, data: bytes):
        self.b, self.i = data, 0

    def uint(self) -> int:
        c = self.b[self.i]
        self.i += 1
        if c < 128:
            return c
        n = 256 - c
        v = int.from_bytes(self.b[self.i: self.i + n], "big")
        self.i += n
        return v

    def int(self) -> int:
        u = self.uint()
        return ~(u >> 1) if u & 1 else u >> 1

    def bytes(self, n: int) -> bytes:
        v = self.b[self.i: self.i + n]
        if len(v) != n:
            raise ValueError("gob: truncated")
        self.i += n
        return v


GOB_UINT, GOB_BYTES, GOB_FIRST_USER = 3, 5, 65
_FIELDS = (("Ccs", GOB_BYTES), ("Pk", GOB_BYTES), ("Vk", GOB_BYTES), ("Curve", GOB_UINT))


def gob_encode_compiled_circuit_bytes(ccs: bytes, pk: bytes, vk: bytes, curve_id: int) -> bytes:
    """What `gob.NewEncoder(&buf).Encode(CompiledCircuitBytes{...})` writes (utils/utils.go:103-112): one type-definition
    message for user type 65, then one value message."""
    name = b"CompiledCircuitBytes"
    td = _gob_int(-GOB_FIRST_USER)
    td += b"\x03"                                                     # wireType.StructT (field 2: delta 3)
    td += b"\x01" + b"\x01" + _gob_uint(len(name)) + name + b"\x01" + _gob_int(GOB_FIRST_USER) + b"\x00"   # CommonType{Name, Id}
    td += b"\x01" + _gob_uint(len(_FIELDS))                            # Field []fieldType
    for fname, fid in _FIELDS:
        td += b"\x01" + _gob_uint(len(fname)) + fname.encode() + b"\x01" + _gob_int(fid) + b"\x00"
    td += b"\x00" + b"\x00"                                            # end structType, end wireType
    val = _gob_int(GOB_FIRST_USER)
    delta = 1
    for v in (ccs, pk, vk):
        if len(v) == 0:                                                # zero values are not transmitted
            delta += 1
            continue
        val += _gob_uint(delta) + _gob_uint(len(v)) + v
        delta = 1
    if curve_id:
        val += _gob_uint(delta) + _gob_uint(curve_id)
    val += b"\x00"
    return _gob_uint(len(td)) + td + _gob_uint(len(val)) + val


def gob_decode_compiled_circuit_bytes(data: bytes) -> Tuple[bytes, bytes, bytes, int]:
    """Inverse of the above; reads the type definition it is given (field names and order come from the stream)."""
    r = _GobReader(data)
    fields: List[Tuple[str, int]] = []
    out = {"Ccs": b"", "Pk": b"", "Vk": b"", "Curve": 0}
    seen_value = False
    while r.i < len(data):
        ln = r.uint()
        end = r.i + ln
        tid = r.int()
        if tid < 0:                                                    # type definition: wireType{StructT: &structType{...}}
            if r.uint() != 3:
                raise ValueError("gob: expected a struct type definition")
            if r.uint() != 1:
                raise ValueError("gob: expected CommonType")
            d = r.uint()
            while d:                                                   # CommonType fields: Name (1), Id (2)
                if d == 1 and not fields and "name" not in out:
                    out["name"] = r.bytes(r.uint()).decode()
                else:
                    r.int()
                d = r.uint()
            if r.uint() != 1:
                raise ValueError("gob: expected the field list")
            for _ in range(r.uint()):
                fname, fid = "", 0
                d = r.uint()
                k = 0
                while d:
                    k += d
                    if k == 1:
                        fname = r.bytes(r.uint()).decode()
                    else:
                        fid = r.int()
                    d = r.uint()
                fields.append((fname, fid))
            r.i = end
            continue
        if not fields:
            raise ValueError("gob: value before its type definition")
        idx = -1
        d = r.uint()
        while d:
            idx += d
            fname, fid = fields[idx]
            if fid == GOB_BYTES:
                out[fname] = r.bytes(r.uint())
            elif fid == GOB_UINT:
                out[fname] = r.uint()
            else:
                raise ValueError("gob: unsupported field type %d" % fid)
            d = r.uint()
        seen_value = True
        r.i = end
    if not seen_value or out.get("name") != "CompiledCircuitBytes":
        raise ValueError("gob: not a CompiledCircuitBytes stream")
    return out["Ccs"], out["Pk"], out["Vk"], out["Curve"]


# ---- gnark-crypto point encodings (compressed) ---------------------------------------------------------------------------
def compress_g1(cv: ecc.ID, P: ecc.Point) -> bytes:
    """G1Affine.Bytes() [UPSTREAM]; flags as observed in the reference's files."""
    n = cv.fp_bytes
    if P is None:
        return bytes([0xC0 if cv is ecc.BLS12_381 else 0x40]) + bytes(n - 1)
    largest = P[1] > (cv.p - 1) // 2
    b = bytearray(P[0].to_bytes(n, "big"))
    if cv is ecc.BLS12_381:
        b[0] |= 0xA0 if largest else 0x80
    else:
        b[0] |= 0xC0 if largest else 0x80
    return bytes(b)


def _g2_decompress(cv: ecc.ID, b: bytes) -> bytes:
    import ctypes as C
    from ._lib import lib, check
    out = C.create_string_buffer(4 * cv.fp_bytes)
    check(lib.apk_g2_decompress(cv.abi, b, out))
    return out.raw


def compress_g2(cv: ecc.ID, raw: bytes) -> bytes:
    """gnark in-memory G2Affine (X.A0 || X.A1 || Y.A0 || Y.A1, Montgomery little-endian) -> compressed X.A1 || X.A0."""
    n = cv.fp_bytes
    rinv = pow(cv.fp_R, -1, cv.p)
    x0, x1, y0, y1 = (int.from_bytes(raw[i * n: (i + 1) * n], "little") * rinv % cv.p for i in range(4))
    if not any(raw):
        return bytes([0xC0 if cv is ecc.BLS12_381 else 0x40]) + bytes(2 * n - 1)
    half = (cv.p - 1) // 2
    largest = (y1 > half) if y1 else (y0 > half)
    b = bytearray(x1.to_bytes(n, "big") + x0.to_bytes(n, "big"))
    if cv is ecc.BLS12_381:
        b[0] |= 0xA0 if largest else 0x80
    else:
        b[0] |= 0xC0 if largest else 0x80
    return bytes(b)


# ---- kzg.ProvingKey / kzg.VerifyingKey (PINNED: the reference's pk.bin / vk.bin) ----------------------------------------------
def write_kzg_pk(cv: ecc.ID, g1: bytes) -> bytes:
    """kzg.ProvingKey.WriteTo: BE u32 count || count compressed G1 (= pk.bin, setup/setup.go:196-228)."""
    pts = cv.g1_vector_decode(g1)
    return struct.pack(">I", len(pts)) + b"".join(compress_g1(cv, P) for P in pts)


def read_kzg_pk(cv: ecc.ID, r: io.BytesIO, device: int = 0) -> bytes:
    """kzg.ProvingKey.ReadFrom; the decompression runs on the GPU (apk_g1_decompress) and validates every point."""
    (count,) = struct.unpack(">I", r.read(4))
    body = r.read(count * cv.fp_bytes)
    if len(body) != count * cv.fp_bytes:
        raise ValueError("kzg proving key: truncated (%d points declared)" % count)
    return setup.decompress_g1_batch(cv, body, device) if count else b""


def write_kzg_vk(cv: ecc.ID, g2: bytes, g1_first: ecc.Point) -> bytes:
    """kzg.VerifyingKey.WriteTo: G2[0] || G2[1] || G1, compressed (= vk.bin; writer order setup/DuskBLS12_381/audit.go:155-179)."""
    w = 4 * cv.fp_bytes
    return compress_g2(cv, g2[:w]) + compress_g2(cv, g2[w: 2 * w]) + compress_g1(cv, g1_first)


def read_kzg_vk(cv: ecc.ID, r: io.BytesIO) -> Tuple[bytes, ecc.Point]:
    w = 2 * cv.fp_bytes
    g2 = _g2_decompress(cv, r.read(w)) + _g2_decompress(cv, r.read(w))
    return g2, setup.decompress_g1(cv, r.read(cv.fp_bytes))


# ---- plonk.VerifyingKey / plonk.ProvingKey WriteTo (UNPINNED: restated from gnark v0.15.0 marshal.go) -----------------------------
def write_plonk_vk(vk: plonk.VerifyingKey) -> bytes:
    """toEncode = Size, SizeInv, Generator, NbPublicVariables, CosetShift, S[0..2], Ql, Qr, Qm, Qo, Qk, Qcp, Kzg.G1, Kzg.G2[0..1],
    CommitmentConstraintIndexes  [UPSTREAM, from memory: unpinned]."""
    cv = vk.curve
    fr = lambda x: (x % cv.r).to_bytes(32, "big")
    g1 = lambda P: compress_g1(cv, P)
    out = struct.pack(">Q", vk.Size) + fr(vk.SizeInv) + fr(vk.Generator) + struct.pack(">Q", vk.NbPublicVariables) + fr(vk.CosetShift)
    out += b"".join(g1(P) for P in vk.S) + b"".join(g1(P) for P in (vk.Ql, vk.Qr, vk.Qm, vk.Qo, vk.Qk))
    out += struct.pack(">I", len(vk.Qcp)) + b"".join(g1(P) for P in vk.Qcp)
    w = 4 * cv.fp_bytes
    out += g1(vk.KzgG1) + compress_g2(cv, vk.KzgG2[:w]) + compress_g2(cv, vk.KzgG2[w: 2 * w])
    out += struct.pack(">I", len(vk.CommitmentConstraintIndexes)) + b"".join(struct.pack(">Q", i) for i in vk.CommitmentConstraintIndexes)
    return out


def read_plonk_vk(cv: ecc.ID, r: io.BytesIO, embedded: bool = False) -> plonk.VerifyingKey:
    """Reads what write_plonk_vk wrote - and nothing else.  embedded = False: the key is everything that is left in `r` (a vk
    file): its length must be exactly this layout's; embedded = True: the key is followed by more data (inside a proving key: the
    kzg.ProvingKey, whose point count must be Size + 3).  A key in any other layout - e.g. one written by gnark, whose list very
    likely carries more fields (the KZG key's precomputed pairing lines) - is refused: the plonk key layout is UNPINNED here
    (tools/gnark_dump + tests/test_gnark_dump.py settle it where Go exists), and nothing unpinned is guessed at."""
    fr = lambda: int.from_bytes(r.read(32), "big")
    g1 = lambda: setup.decompress_g1(cv, r.read(cv.fp_bytes))
    (size,) = struct.unpack(">Q", r.read(8))
    size_inv, gen = fr(), fr()
    (nbp,) = struct.unpack(">Q", r.read(8))
    shift = fr()
    S = [g1() for _ in range(3)]
    ql, qr, qm, qo, qk = (g1() for _ in range(5))
    (nq,) = struct.unpack(">I", r.read(4))
    qcp = [g1() for _ in range(nq)]
    kg1 = g1()
    w = 2 * cv.fp_bytes
    g2 = _g2_decompress(cv, r.read(w)) + _g2_decompress(cv, r.read(w))
    here = r.tell()
    tail = 4 + 8 * nq                                     # CommitmentConstraintIndexes: u32 count + count x u64, one index per Qcp
    head = r.read(4)
    ok = len(head) == 4 and struct.unpack(">I", head)[0] == nq
    if ok and not embedded:
        ok = len(r.getbuffer()) - here == tail
    if ok and embedded:
        r.seek(here + tail)
        nxt = r.read(4)                                   # kzg.ProvingKey: its G1 count is Size + 3 (setup/setup.go:113-114)
        ok = len(nxt) == 4 and struct.unpack(">I", nxt)[0] == size + 3
    if not ok:
        raise ValueError("plonk verifying key: what follows Kzg.G2 is not this package's layout (a CommitmentConstraintIndexes list of %d "
                         "entries%s): the plonk key layout here is unpinned and keys written elsewhere are not read"
                         % (nq, " and then a kzg proving key of %d points" % (size + 3) if embedded else ", then the end of the key"))
    r.seek(here + 4)
    cci = [struct.unpack(">Q", r.read(8))[0] for _ in range(nq)]
    if size == 0 or size & (size - 1) or size_inv * size % cv.r != 1 or pow(gen, size, cv.r) != 1:
        raise ValueError("plonk verifying key: inconsistent domain fields")
    return plonk.VerifyingKey(curve=cv, Size=size, SizeInv=size_inv, Generator=gen, CosetShift=shift, NbPublicVariables=nbp, Ql=ql, Qr=qr,
                              Qm=qm, Qo=qo, Qk=qk, S=S, Qcp=qcp, CommitmentConstraintIndexes=cci, KzgG1=kg1, tau=None, KzgG2=g2)


def write_plonk_pk(vk: plonk.VerifyingKey, srs: setup.SRS) -> bytes:
    """plonk.ProvingKey.WriteTo (gnark >= 0.9: {Kzg, KzgLagrange, Vk}): Vk || Kzg || KzgLagrange [UPSTREAM: unpinned]."""
    return write_plonk_vk(vk) + write_kzg_pk(srs.curve, srs.g1) + write_kzg_pk(srs.curve, srs.g1_lagrange or b"")


def read_plonk_pk(cv: ecc.ID, r: io.BytesIO, device: int = 0) -> Tuple[plonk.VerifyingKey, setup.SRS]:
    vk = read_plonk_vk(cv, r, embedded=True)
    g1 = read_kzg_pk(cv, r, device)
    lag = read_kzg_pk(cv, r, device)
    return vk, setup.SRS(cv, vk.Size, g1, lag or None, None, vk.KzgG2)


# ---- constraint system ------------------------------------------------------------------------------------------------
CCS_TAG = b"APKCCS1\n"   # this package's own encoding; gnark's ccs.WriteTo is CBOR and starts differently


def _closure_free_solver(ccs: frontend.ConstraintSystem):
    """The frontend's solver entries for API-built circuits are Python closures; every one of them restates the constraint
    emitted next to it, so the storable form is recovered from the constraints: (ql, qr, qm, -1, qk, a, b, c) defines wire c,
    (0, 0, 1, 0, -1, a, inv, 0) defines the inverse wire."""
    if isinstance(ccs.solver, str):
        return ccs.solver
    r = ccs.field
    by_out, by_inv = {}, {}
    for ql, qr, qm, qo, qk, xa, xb, xc in ccs.constraints:
        if qo == r - 1:
            by_out.setdefault(xc, ("gate", ql, qr, qm, qk, xa, xb))
        elif (ql, qr, qm, qo, qk) == (0, 0, 1, 0, r - 1):
            by_inv.setdefault(xb, ("inv", xa))
    out = []
    for wire, fn in ccs.solver:
        if isinstance(fn, tuple):
            out.append([wire, list(fn)])
        elif wire in by_out:
            out.append([wire, list(by_out[wire])])
        elif wire in by_inv:
            out.append([wire, list(by_inv[wire])])
        else:
            raise ValueError("wire %d has no defining constraint: cannot serialise its solver step" % wire)
    return out


def write_ccs(ccs: frontend.ConstraintSystem) -> bytes:
    return CCS_TAG + json.dumps({"r": ccs.field, "public": ccs.public_names, "secret": ccs.secret_names, "constraints": [list(c) for c in ccs.constraints],
                                 "solver": _closure_free_solver(ccs), "nb_variables": ccs.nb_variables,
                                 "commitments": [[list(rows), cidx] for rows, cidx in ccs.commitments]}).encode()


def read_ccs(b: bytes) -> frontend.ConstraintSystem:
    if not b.startswith(CCS_TAG):
        raise ValueError("constraint system is gnark CBOR (ccs.WriteTo): not parsed here - tools/gnark_dump exports the trace "
                         "(Ql..Qk, S) and the solved wires, which is what libapk consumes (include/apk.h apk_circuit_desc)")
    d = json.loads(b[len(CCS_TAG):])
    solver = d["solver"] if isinstance(d["solver"], str) else [(s[0], tuple(s[1])) for s in d["solver"]]
    return frontend.ConstraintSystem(d["r"], d["public"], d["secret"], [tuple(c) for c in d["constraints"]], solver, d["nb_variables"],
                                     [(list(rows), cidx) for rows, cidx in d["commitments"]])


# ---- utils.SerializeCompiledCircuit / DeserializeCompiledCircuit -----------------------------------------------------------
@dataclass
class CompiledCircuitBytes:
    """utils/utils.go:88-94."""
    Ccs: bytes
    Pk: bytes
    Vk: bytes
    Curve: int


def SerializeCompiledCircuit(cc, srs: setup.SRS, filepath: str) -> None:
    """utils.SerializeCompiledCircuit (utils/utils.go:97-122).  `srs` is the SRS the circuit was set up with (gnark keeps it
    inside its ProvingKey; here the proving key is a device context, so the caller hands the SRS over)."""
    blob = gob_encode_compiled_circuit_bytes(write_ccs(cc.Ccs), write_plonk_pk(cc.Vk, srs), write_plonk_vk(cc.Vk), ECC_ID[cc.Curve.name])
    with open(filepath, "wb") as f:
        f.write(blob)


def DeserializeCompiledCircuit(filepath: str, device: int = 0, slots: int = 1):
    """utils.DeserializeCompiledCircuit (utils/utils.go:124-157): rebuilds the device context (plonk.Setup's GPU part) from
    the stored SRS and constraint system and checks that the stored verifying key is the one that setup derives."""
    from .algoplonk import CompiledCircuit
    with open(filepath, "rb") as f:
        data = f.read()
    try:
        ccs_b, pk_b, vk_b, cid = gob_decode_compiled_circuit_bytes(data)
    except Exception as e:
        raise ValueError("error decoding compiled circuit: %s" % e)
    cv = ECC_BY_ID.get(cid)
    if cv is None:
        raise ValueError("error decoding compiled circuit: unsupported curve id %d" % cid)
    try:
        ccs = read_ccs(ccs_b)
    except Exception as e:
        raise ValueError("error reading CCS data: %s" % e)
    try:
        vk_in_pk, srs = read_plonk_pk(cv, io.BytesIO(pk_b), device)
    except Exception as e:
        raise ValueError("error reading PK data: %s" % e)
    try:
        vr = io.BytesIO(vk_b)
        vk = read_plonk_vk(cv, vr)
        if vr.read(1):
            raise ValueError("%d unconsumed byte(s) after the verifying key - a gnark-written key carries fields this reader does not "
                             "know (e.g. Kzg.Lines): the plonk key layout here is unpinned" % (len(vk_b) - vr.tell() + 1))
    except Exception as e:
        raise ValueError("error reading VK data: %s" % e)
    pk, vk_now = plonk.Setup(ccs, srs, device=device, slots=slots)
    for a, b in ((vk, vk_now), (vk_in_pk, vk_now)):
        if (a.Ql, a.Qr, a.Qm, a.Qo, a.Qk, a.S, a.Qcp, a.KzgG1, a.KzgG2) != (b.Ql, b.Qr, b.Qm, b.Qo, b.Qk, b.S, b.Qcp, b.KzgG1, b.KzgG2):
            raise ValueError("error reading VK data: stored verifying key does not match the circuit and SRS")
    return CompiledCircuit(ccs, pk, vk_now, cv)
