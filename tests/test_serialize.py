"""Row f4 of SURVEY.md section 8: the byte formats behind utils.SerializeCompiledCircuit / DeserializeCompiledCircuit
(/root/reference/utils/utils.go:88-157)."""
import io
import os

import pytest

from algoplonk_amd import ecc, frontend, plonk as ap_plonk, serialize as ser, setup as ap_setup
from oracle import curves as oc

from helpers import CURVES

G = os.path.join(os.path.dirname(__file__), "golden")


def test_gob_primitives_match_the_gob_specification_example():
    """The worked example of Go's encoding/gob documentation: `type Point struct{X, Y int}` with X = 22, Y = 33 encodes as
    1f ff 81 03 01 01 05 50 6f 69 6e 74 01 ff 82 00 01 02 01 01 58 01 04 00 01 01 59 01 04 00 00 00 | 07 ff 82 01 2c 01 42 00.
    The same primitives build the CompiledCircuitBytes envelope."""
    assert ser._gob_int(-65) == bytes.fromhex("ff81") and ser._gob_int(65) == bytes.fromhex("ff82")
    assert ser._gob_int(22) == b"\x2c" and ser._gob_int(33) == b"\x42" and ser._gob_int(2) == b"\x04"
    assert ser._gob_uint(7) == b"\x07" and ser._gob_uint(256) == bytes.fromhex("fe0100") and ser._gob_uint(128) == bytes.fromhex("ff80")
    r = ser._GobReader(bytes.fromhex("ff81" "ff82" "2c" "fe0100"))
    assert (r.int(), r.int(), r.int(), r.uint()) == (-65, 65, 22, 256)


def test_gob_envelope_round_trip_and_layout():
    ccs, pk, vk = b"c" * 300, os.urandom(1000), b"\x00\x01\x02"
    blob = ser.gob_encode_compiled_circuit_bytes(ccs, pk, vk, 3)
    # type definition message: id -65, struct "CompiledCircuitBytes" with fields Ccs, Pk, Vk ([]byte = 5 -> 0a) and Curve (uint = 3 -> 06)
    assert blob[1:3] == bytes.fromhex("ff81") and b"\x14CompiledCircuitBytes" in blob[:40]
    for name, tid in ((b"\x03Ccs", 0x0A), (b"\x02Pk", 0x0A), (b"\x02Vk", 0x0A), (b"\x05Curve", 0x06)):
        i = blob.index(name)
        assert blob[i + len(name): i + len(name) + 3] == bytes([0x01, tid, 0x00])
    assert ser.gob_decode_compiled_circuit_bytes(blob) == (ccs, pk, vk, 3)
    # zero-valued fields are not transmitted (gob): an empty Pk shifts the next field's delta
    assert ser.gob_decode_compiled_circuit_bytes(ser.gob_encode_compiled_circuit_bytes(ccs, b"", vk, 1)) == (ccs, b"", vk, 1)
    with pytest.raises(ValueError):
        ser.gob_decode_compiled_circuit_bytes(blob[:-4])


@pytest.mark.parametrize("name,cname", [("PerpetualPowersOfTauBN254", "bn254"), ("EethereumKzgCeremonyBLS12_381", "bls12-381"), ("DuskBLS12_381", "bls12-381")])
def test_reference_vk_bin_files_through_the_kzg_reader_and_writer(name, cname):
    """PINNED layer: the reference's vk.bin files are kzg.VerifyingKey.WriteTo outputs (setup/setup.go:174,190): parse them, write
    them back, get the same bytes (the compressor must agree with gnark's on every flag bit)."""
    cv, ov = CURVES[cname]
    vkb = open(os.path.join(G, name + ".vk.bin"), "rb").read()
    g2, g1 = ser.read_kzg_vk(cv, io.BytesIO(vkb))
    assert g1 == ov.g1
    assert ser.write_kzg_vk(cv, g2, g1) == vkb


def test_reference_pk_bin_head_through_the_point_compressor():
    """PINNED layer: the head of the Ethereum ceremony's pk.bin = kzg.ProvingKey.WriteTo (count || compressed points): the
    oracle decompresses, this module's writer must reproduce the file bytes."""
    cv, ov = CURVES["bls12-381"]
    head = open(os.path.join(G, "EethereumKzgCeremonyBLS12_381.pk.head.bin"), "rb").read()
    pts = [ov.decompress(head[4 + 48 * i: 4 + 48 * (i + 1)]) for i in range(8)]
    assert ser.write_kzg_pk(cv, cv.g1_vector(pts))[4:] == head[4:]
    assert ser.compress_g1(cv, None)[0] == 0xC0 and ser.compress_g1(ecc.BN254, None)[0] == 0x40


@pytest.mark.parametrize("cname", ["bn254", "bls12-381"])
def test_plonk_vk_and_ccs_round_trip(cname):
    """UNPINNED layer (field order of gnark's plonk VerifyingKey.WriteTo restated from memory): self-consistency only."""
    from oracle import circuits as ocircuits, plonk as oplonk
    from oracle.prng import tau_from_seed
    cv, ov = CURVES[cname]
    c, sol, plan = ocircuits.bsb22_square(ov, 1)
    tau = tau_from_seed(3, cv.r)
    ovk = oplonk.setup(c, oplonk.synthetic_srs(ov, c.domain_size(), tau, materialize=False)).vk
    vk = ap_plonk.VerifyingKey(curve=cv, Size=ovk.size, SizeInv=ovk.size_inv, Generator=ovk.generator, CosetShift=ovk.coset_shift,
                               NbPublicVariables=ovk.nb_public, Ql=ovk.ql, Qr=ovk.qr, Qm=ovk.qm, Qo=ovk.qo, Qk=ovk.qk, S=list(ovk.s), Qcp=list(ovk.qcp),
                               CommitmentConstraintIndexes=list(ovk.commitment_constraint_indexes), KzgG1=ovk.g1, tau=None,
                               KzgG2=ap_setup.g2_from_tau(cv, tau))
    b = ser.write_plonk_vk(vk)
    pt = cv.fp_bytes
    assert len(b) == 8 + 32 + 32 + 8 + 32 + 8 * pt + 4 + pt + pt + 4 * pt + 4 + 8
    back = ser.read_plonk_vk(cv, io.BytesIO(b))
    assert back == vk
    with pytest.raises(ValueError, match="inconsistent"):
        ser.read_plonk_vk(cv, io.BytesIO(b[:7] + bytes([b[7] ^ 3]) + b[8:]))
    # only this package's own layout is read: anything between Kzg.G2[1] and the index list (a key written by gnark very likely
    # carries the KZG key's pairing lines there) is refused, not guessed at
    tail = 4 + 8 * len(vk.CommitmentConstraintIndexes)
    with pytest.raises(ValueError, match="unpinned"):
        ser.read_plonk_vk(cv, io.BytesIO(b[:-tail] + bytes(77) + b[-tail:]))

    class Sq(frontend.Circuit):
        X = frontend.Public(); Y = frontend.Secret()

        def define(self, api):
            api.AssertIsEqual(self.X, api.Mul(self.Y, self.Y))
            api.AssertIsDifferent(api.Commit(self.Y, self.X))

    ccs = frontend.Compile(cv.r, Sq())
    ccs2 = ser.read_ccs(ser.write_ccs(ccs))
    assert (ccs2.constraints, ccs2.commitments, ccs2.nb_variables, ccs2.public_names) == (ccs.constraints, ccs.commitments, ccs.nb_variables, ccs.public_names)
    a = Sq(); a.X, a.Y = 9, 3
    w = frontend.NewWitness(a, cv.r)
    hint = lambda col: 77
    assert frontend.solve(ccs2, w, hint, [(1, 2)], []) == frontend.solve(ccs, w, hint, [(1, 2)], [])
    with pytest.raises(ValueError, match="gnark CBOR"):
        ser.read_ccs(b"\xa5\x01\x02")


@pytest.mark.gpu
@pytest.mark.parametrize("cname", ["bn254", "bls12-381"])
def test_serialize_deserialize_compiled_circuit_then_prove(gpu, cname, tmp_path):
    """utils.SerializeCompiledCircuit -> DeserializeCompiledCircuit (utils/utils.go:97-157) -> the restored circuit proves and
    verifies, and its proof equals the original's for the same blinding."""
    from algoplonk_amd import Compile, MarshalProof
    cv, ov = CURVES[cname]

    class Pyth(frontend.Circuit):
        A = frontend.Public(); B = frontend.Public(); C = frontend.Secret()

        def define(self, api):
            api.AssertIsEqual(api.Add(api.Mul(self.A, self.A), api.Mul(self.B, self.B)), api.Mul(self.C, self.C))

    name = ap_setup.TestOnlySetup(cv)
    seed = 0x5E71A1
    cc = Compile(Pyth(), cv, name, device=gpu, seed=seed)
    tau = int.from_bytes(seed.to_bytes(48, "big"), "big") % cv.r
    srs = ap_setup.unsafe_srs(cv, 8, tau, device=gpu, lagrange=True)
    path = str(tmp_path / "cc.gob")
    ser.SerializeCompiledCircuit(cc, srs, path)
    cc2 = ser.DeserializeCompiledCircuit(path, device=gpu)
    assert cc2.Vk.Ql == cc.Vk.Ql and cc2.Vk.S == cc.Vk.S and cc2.Curve is cv
    a = Pyth(); a.A, a.B, a.C = 3, 4, 5
    bl = list(range(101, 110))
    assert MarshalProof(cc2.Verify(a, blinding=bl).Proof) == MarshalProof(cc.Verify(a, blinding=bl).Proof)
    bad = bytearray(open(path, "rb").read()); bad[len(bad) // 2] ^= 1
    open(path, "wb").write(bytes(bad))
    with pytest.raises(ValueError):
        ser.DeserializeCompiledCircuit(path, device=gpu)
    cc.Pk.close(); cc2.Pk.close()


def test_plonk_vk_inside_a_proving_key_picks_its_layout_from_what_follows():
    """inside a proving key the verifying key is followed by the kzg proving key, whose point count must be Size + 3 -
    that, not four zero bytes, decides whether the layout is the one this package writes (a key without commitments has nq = 0, and
    any four zero bytes used to pass for its empty index list)."""
    import struct
    from oracle import circuits as ocircuits, plonk as oplonk
    from oracle.prng import tau_from_seed
    cv, ov = CURVES["bn254"]
    c, sol = ocircuits.pythagorean(ov) if hasattr(ocircuits, "pythagorean") else (None, None)
    if c is None:
        c, sol, _ = ocircuits.bsb22_square(ov, 0)
    tau = tau_from_seed(3, cv.r)
    ovk = oplonk.setup(c, oplonk.synthetic_srs(ov, c.domain_size(), tau, materialize=False)).vk
    vk = ap_plonk.VerifyingKey(curve=cv, Size=ovk.size, SizeInv=ovk.size_inv, Generator=ovk.generator, CosetShift=ovk.coset_shift,
                               NbPublicVariables=ovk.nb_public, Ql=ovk.ql, Qr=ovk.qr, Qm=ovk.qm, Qo=ovk.qo, Qk=ovk.qk, S=list(ovk.s), Qcp=[],
                               CommitmentConstraintIndexes=[], KzgG1=ovk.g1, tau=None, KzgG2=ap_setup.g2_from_tau(cv, tau))
    b = ser.write_plonk_vk(vk)
    follow = struct.pack(">I", vk.Size + 3) + bytes(40)            # the head of a kzg proving key: its G1 count
    assert ser.read_plonk_vk(cv, io.BytesIO(b + follow), embedded=True) == vk
    with pytest.raises(ValueError, match="unpinned"):     # a block of anything in front of the (empty) index list: refused
        ser.read_plonk_vk(cv, io.BytesIO(b[:-4] + bytes(4) + bytes(range(1, 200)) + b[-4:] + follow), embedded=True)
    with pytest.raises(ValueError, match="unpinned"):
        ser.read_plonk_vk(cv, io.BytesIO(b + struct.pack(">I", vk.Size + 4) + bytes(40)), embedded=True)
    # a vk FILE: the layout follows from the remaining length alone
    with pytest.raises(ValueError, match="unpinned"):
        ser.read_plonk_vk(cv, io.BytesIO(b + bytes(4)))
