"""The AlgoPlonk public API, mirrored name for name from /root/reference/algoplonk.go and helper.go:

    Compile(circuit, curve, setupConfig) -> CompiledCircuit          algoplonk.go:37-59
    CompiledCircuit.Verify(assignment)   -> VerifiedProof            algoplonk.go:79-98  (calls plonk.Prove :89)
    VerifiedProof.ExportProofAndPublicInputs / WriteProof / WritePublicInputs      algoplonk.go:103-156
    MarshalProof / MarshalPublicInputs                                helper.go:13-24, 91-110

Out of scope here: WritePuyaPyVerifier and the AVM tooling.  `plonk.Verify` (algoplonk.go:93) is
gnark's pairing verifier on the Go side; here it is `algoplonk_amd.plonk.Verify` -> libapk's host-side apk_verify, and
it ALWAYS runs: a `VerifiedProof` is never handed out unverified.  The optional `verifier` callable runs in addition
(the test-suite plugs in the verifier it transcribed from the reference's templates).
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Callable, Optional, Sequence

from . import ecc, frontend, plonk, setup
from . import _lib
from ._lib import lib, check


@dataclass
class CompiledCircuit:
    Ccs: frontend.ConstraintSystem
    Pk: plonk.ProvingKey
    Vk: plonk.VerifyingKey
    Curve: ecc.ID

    def Verify(self, assignment: frontend.Circuit, blinding: Optional[Sequence[int]] = None,
               verifier: Optional[Callable[[plonk.VerifyingKey, bytes, bytes], bool]] = None) -> "VerifiedProof":
        try:
            witness = frontend.NewWitness(assignment, self.Curve.ScalarField())
        except Exception as e:
            raise ValueError("error creating witness: %s" % e)
        try:
            proof = plonk.Prove(self.Ccs, self.Pk, witness, blinding)
        except Exception as e:
            raise RuntimeError("error creating Plonk proof: %s" % e)
        try:
            plonk.Verify(proof, self.Vk, witness.Public())                     # algoplonk.go:93
        except Exception as e:
            raise RuntimeError("error verifying Plonk proof: %s" % e)
        if verifier is not None and not verifier(self.Vk, MarshalProof(proof), MarshalPublicInputs(witness)):
            raise RuntimeError("error verifying Plonk proof: rejected by the caller's verifier")
        return VerifiedProof(proof, witness)


@dataclass
class VerifiedProof:
    Proof: plonk.Proof
    Witness: frontend.Witness

    def ExportProofAndPublicInputs(self, proofFilePath: str, publicInputsFilePath: str) -> None:
        if proofFilePath:
            with open(proofFilePath, "wb") as f:
                self.WriteProof(f)
        if publicInputsFilePath:
            with open(publicInputsFilePath, "wb") as f:
                self.WritePublicInputs(f)

    def WriteProof(self, w) -> None:
        w.write(MarshalProof(self.Proof))

    def WritePublicInputs(self, w) -> None:
        w.write(MarshalPublicInputs(self.Witness))


def Compile(circuit: frontend.Circuit, curve: ecc.ID, setupConfig, device: int = 0, seed: Optional[int] = None,
            msm_window: int = 0) -> CompiledCircuit:
    if curve is not ecc.BN254 and curve is not ecc.BLS12_381:
        raise ValueError("unsupported curve: %s" % curve)
    setupInfo, ok = setup.Get(setupConfig)
    if not ok:
        raise ValueError("unknown setup: %s" % (setupConfig,))
    if curve is not setupInfo.Curve:
        raise ValueError("setup curve %s does not match circuit curve %s" % (setupInfo.Curve, curve))
    try:
        ccs = frontend.Compile(curve.ScalarField(), circuit)
    except Exception as e:
        raise ValueError("error compiling circuit: %s" % e)
    try:
        pk, vk = Run(ccs, setupConfig, device=device, seed=seed, msm_window=msm_window)
    except Exception as e:
        raise RuntimeError("error setting up Plonk: %s" % e)
    return CompiledCircuit(ccs, pk, vk, curve)


def Run(ccs: frontend.ConstraintSystem, setupConfig, device: int = 0, seed: Optional[int] = None, msm_window: int = 0,
        slots: int = 1):
    """setup.Run(ccs, setupConfig) (setup/setup.go:95-150)."""
    info, ok = setup.Get(setupConfig)
    if not ok:
        raise ValueError("unknown setup: %s" % (setupConfig,))
    n = ccs.domain_size()
    if not info.Trusted:
        tau = int.from_bytes(os.urandom(48) if seed is None else seed.to_bytes(48, "big"), "big") % info.Curve.r
        if tau < 2:
            tau += 2
        srs = setup.unsafe_srs(info.Curve, n, tau, device=device, lagrange=bool(ccs.commitments))
    else:
        srs = setup.trusted_srs(info, n, device=device, lagrange=bool(ccs.commitments))
    return plonk.Setup(ccs, srs, device=device, msm_window=msm_window, slots=slots)


def MarshalProof(proof: plonk.Proof) -> bytes:
    """helper.go:13-24: BN254 = gnark MarshalSolidity layout, BLS12-381 = helper.go:27-88."""
    if not isinstance(proof, plonk.Proof):
        raise TypeError("unrecognized proof type")
    cap = 9 * 96 + 6 * 32 + _lib.MAX_COMMITMENTS * (32 + 96)
    out = C.create_string_buffer(cap)
    n = C.c_size_t(0)
    check(lib.apk_marshal_proof(C.byref(proof.raw), out, cap, C.byref(n)))
    return out.raw[: n.value]


def MarshalPublicInputs(witness: frontend.Witness) -> bytes:
    """helper.go:91-110: the public part of the witness, 32 big-endian bytes per input."""
    pub = witness.Public().public
    cv = ecc.BN254 if witness.field == ecc.BN254.r else ecc.BLS12_381
    out = C.create_string_buffer(max(32 * len(pub), 1))
    check(lib.apk_marshal_public_inputs(cv.abi, cv.fr_vector(pub), len(pub), out, 32 * len(pub)))
    return out.raw[: 32 * len(pub)]
