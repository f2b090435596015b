"""algoplonk_amd - MI355X-native PLONK prover path for AlgoPlonk.

The product is libapk.so (HIP kernels for gfx950 + a C-ABI, include/apk.h).  This package is the host-side
mirror of the reference's Go API for the path (algoplonk.go / helper.go / setup/setup.go) over that C-ABI.
"""
from . import ecc, frontend, setup, plonk  # noqa: F401
from .algoplonk import (Compile, CompiledCircuit, VerifiedProof, MarshalProof, MarshalPublicInputs, Run)  # noqa: F401

__all__ = ["ecc", "frontend", "setup", "plonk", "Compile", "CompiledCircuit", "VerifiedProof", "MarshalProof",
           "MarshalPublicInputs", "Run"]
