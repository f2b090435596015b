"""CPU tier: the two tools the parity pin rests on (tests/golden/make_template_fixtures.py), tested on their own so that a
verdict in tests/golden/template_verdicts.json is the reference template's and not an artefact of the harness:
the text/template subset that renders the reference's verifier templates, and the algopy / AVM shim that executes them.
(The generator itself reads /root/reference and runs in the build container only; nothing here does.)"""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
import make_template_fixtures as m   # noqa: E402


class _NS:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def _render(text, data, **funcs):
    fm = {"inc": lambda i: i + 1, "add": lambda a, b: a + b, "mul": lambda a, b: a * b}
    fm.update(funcs)
    return m.GoTemplate(text, fm).render(data)


def test_text_template_subset_follows_go_semantics():
    d = _NS(L=[10, 20, 30], N=2, Inner=_NS(X=_NS(A0=7)), Empty=[])
    # range with index and element, dot rebound to the element, $ = the root, nested field access
    assert _render("{{ range $i, $e := .L }}[{{ $i }}:{{ $e }}:{{ $.N }}]{{ end }}", d) == "[0:10:2][1:20:2][2:30:2]"
    assert _render("{{ .Inner.X.A0 }}", d) == "7"
    # parenthesised calls nest; len and gt are builtins
    assert _render("{{ add 24 (mul 3 (len .L)) }}", d) == "33"
    assert _render("{{ if gt (len .L) 0 }}yes{{ end }}{{ if gt (len .Empty) 0 }}no{{ end }}", d) == "yes"
    assert _render("{{ range $i, $e := .L }}{{ inc $i }}{{ end }}", d) == "123"
    # trim markers: `{{- ` eats the white space before, ` -}}` the white space after, including new lines
    assert _render("a \n {{- .N }}  b", d) == "a2  b"
    assert _render("a  {{ .N -}} \n\t b", d) == "a  2b"
    assert _render("x\n{{ range $i, $e := .L -}}\n  v{{ $i }}\n{{ end -}}\ny", d) == "x\nv0\nv1\nv2\ny"
    # an empty range renders nothing; an action with a function of no arguments
    assert _render("<{{ range $i, $e := .Empty }}never{{ end }}>", d) == "<>"
    assert _render("{{ (name) }}", d, name=lambda: "Verifier") == "Verifier"
    # a comment may contain `}}` (the reference wraps a py.log call in one: templateSmartContractBN254.go:133)
    assert _render('a{{/*}}py.log("x"){{*/ -}}\n  b', d) == "ab"
    with pytest.raises(AssertionError):
        m.GoTemplate("{{ range $i, $e := .L }}unbalanced", {})


def test_avm_shim_byte_math_follows_the_avm():
    B, Y = m.BigUInt, m.Bytes
    # byte-math results carry no leading zeros; zero is the empty string; b| left-pads the shorter operand
    assert B(0).bytes == b"" and B(256).bytes == b"\x01\x00" and B.from_bytes(b"\x00\x00\x05") == 5
    assert (Y(bytes(32)) | B(5).bytes) == bytes(31) + b"\x05"
    assert (Y(b"\x0f\xf0") ^ Y(b"\xff")) == b"\x0f\x0f"
    # no negative results, no division by zero, operands of at most 64 bytes
    with pytest.raises(m.AvmError):
        B(3) - B(4)
    with pytest.raises(m.AvmError):
        B(3) % B(0)
    with pytest.raises(m.AvmError):
        B(1 << 512) * B(2)
    assert (B(1 << 511) * B(2)).v == 1 << 512            # the RESULT may be longer than 64 bytes
    assert (B(7) // 2).v == 3 and (B(7) % 2) == 1 and B(5) >= 5 and not B(5) > 5
    # arc4.UInt256: fixed 32 bytes, encoding fails on overflow
    assert m.UInt256(B(1)).bytes == bytes(31) + b"\x01"
    with pytest.raises(m.AvmError):
        m.UInt256(1 << 256)
    # slices clamp like the AVM's substring helpers in puya; indexing out of range fails
    assert Y(b"abcdef")[2:4] == b"cd" and Y(b"abcdef")[4:99] == b"ef" and Y(b"abc")[2:] == b"c" and Y(b"abc").length == 3
    # uint64 does not wrap
    with pytest.raises(m.AvmError):
        m.UInt64(0) - 1
    assert [int(i) for i in m.urange(m.UInt64(3))] == [0, 1, 2]
    # DynamicArray bounds
    arr = m._DynamicArray()
    arr.append(m.UInt256(9))
    assert arr[m.UInt64(0)].bytes[31:] == b"\x09"
    with pytest.raises(m.AvmError):
        arr[1]


def test_avm_shim_ec_ops_and_helpers():
    from oracle import curves as oc
    for ov in (oc.BN254, oc.BLS12_381):
        m.ensure_mul_raw(ov)
        pr = m.pairing_module(ov)
        avm = m.Avm(ov, pr)
        m.install_algopy(avm)
        from algopy.op import EllipticCurve as ec, EC, bzero, setbit_bytes, sha256
        tag = EC.BN254g1 if ov.name == "bn254" else EC.BLS12_381g1
        n = ov.fp_bytes
        G = avm.g1_out(ov.g1)
        inf = bzero(2 * n)
        # infinity is the all-zero string for the ec ops (what the BN254 template therefore also hashes)
        assert ec.add(tag, G, inf) == G and ec.scalar_mul(tag, G, m.BigUInt(0).bytes) == inf
        assert ec.scalar_mul(tag, G, m.BigUInt(5).bytes) == avm.g1_out(ov.mul(ov.g1, 5))
        # 0x40 in the first byte is NOT a point for the ec ops; a point off the curve fails the program; scalars are at most 32 bytes
        for bad in (bytes([0x40]) + bytes(2 * n - 1), G.b[:-1] + bytes([G.b[-1] ^ 1])):
            with pytest.raises(m.AvmError):
                ec.add(tag, m.Bytes(bad), G)
        with pytest.raises(m.AvmError):
            ec.scalar_mul(tag, G, bytes(33))
        with pytest.raises(m.AvmError):
            ec.add(EC.BN254g1 if ov.name != "bn254" else EC.BLS12_381g1, G, G)
        # setbit_bytes counts bits from the most significant bit of byte 0 (the BLS template's fs(): 0x80, not gnark's 0x40)
        assert setbit_bytes(bzero(4), 0, True) == b"\x80\x00\x00\x00" and setbit_bytes(bzero(2), 15, True) == b"\x00\x01"
        assert sha256(m.Bytes(b"abc")).b.hex().startswith("ba7816bf")
        # e(aG1, G2) e(-G1, aG2) = 1 through the shim's pairing_check (G2 = X.A0 || X.A1 || Y.A0 || Y.A1)
        a = 0xC0FFEE
        enc2 = lambda Q: b"".join(c.to_bytes(n, "big") for c in (Q[0][0], Q[0][1], Q[1][0], Q[1][1]))
        g1s = avm.g1_out(ov.mul(ov.g1, a)).b + avm.g1_out(ov.neg(ov.g1)).b
        g2s = enc2(pr.G2_GEN) + enc2(pr.g2_mul(pr.G2_GEN, a))
        assert ec.pairing_check(tag, m.Bytes(g1s), m.Bytes(g2s)) is True
        assert ec.pairing_check(tag, m.Bytes(g1s), m.Bytes(enc2(pr.G2_GEN) + enc2(pr.g2_mul(pr.G2_GEN, a + 1)))) is False


def test_the_fixture_keeps_the_finding_as_a_negative_control():
    """Rounds 1-2 hashed infinity as 0x40 00.. on both curves; the executed BN254 template rejects a proof made that way."""
    import json
    fix = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "template_verdicts.json")))
    nc = fix["negative_control"]
    assert nc["verdict"] == "reject" and "0x40" in nc["what"]
    from oracle import curves as oc, plonk as oplonk
    # and the oracle's verifier, which now encodes BN254 infinity as the template does, rejects it as well
    case = next(c for c in fix["cases"] if c["curve"] == "bn254" and c["circuit"] == "pythagorean")
    sys.path.insert(0, os.path.dirname(__file__))
    from test_template_pin import _oracle_vk
    assert not oplonk.verify(_oracle_vk(oc.BN254, case["vk"]), bytes.fromhex(nc["proof"]), bytes.fromhex(nc["public_inputs"]))
