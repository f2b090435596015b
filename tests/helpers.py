"""Shared plumbing for the parity tests: build the same circuit for the oracle and for the HIP path."""
from __future__ import annotations

from typing import List, Tuple

from algoplonk_amd import ecc, frontend, plonk as ap_plonk, setup as ap_setup
from oracle import curves as ocurves, plonk as oplonk
from oracle.prng import SplitMix64

CURVES = {"bn254": (ecc.BN254, ocurves.BN254), "bls12-381": (ecc.BLS12_381, ocurves.BLS12_381)}


def oracle_circuit_from_ccs(ocv, ccs: frontend.ConstraintSystem) -> oplonk.Circuit:
    return oplonk.Circuit(ocv, ccs.GetNbPublicVariables(), ccs.nb_variables, list(ccs.constraints))


def blinding(cv, seed: int) -> List[int]:
    g = SplitMix64(seed)
    return [g.fr(cv.r) for _ in range(9)]


def oracle_vk_from_product(ovk_curve, vk: ap_plonk.VerifyingKey) -> oplonk.VerifyingKey:
    """Feed the verifying key the HIP path produced to the oracle's transcribed verifier."""
    return oplonk.VerifyingKey(
        curve=ovk_curve, size=vk.Size, size_inv=vk.SizeInv, generator=vk.Generator, coset_shift=vk.CosetShift,
        nb_public=vk.NbPublicVariables, ql=vk.Ql, qr=vk.Qr, qm=vk.Qm, qo=vk.Qo, qk=vk.Qk, s=list(vk.S), qcp=list(vk.Qcp),
        commitment_constraint_indexes=list(vk.CommitmentConstraintIndexes), g1=vk.KzgG1, tau=vk.tau)


def random_chain_ccs(cv, log_n: int, seed: int, nb_public: int = 2) -> Tuple[frontend.ConstraintSystem, frontend.Witness, List[int]]:
    """BASELINE.md §2 random circuit, built directly as a ConstraintSystem (no per-gate closures)."""
    r = cv.r
    n = 1 << log_n
    g = SplitMix64(seed)
    sol = [g.fr(r) for _ in range(nb_public + 2)]
    cons = []
    for _ in range(n - nb_public):
        nv = len(sol)
        xa, xb = g.below(nv), g.below(nv)
        ql, qr, qm, qk = g.fr(r), g.fr(r), g.fr(r), g.fr(r)
        a, b = sol[xa], sol[xb]
        cons.append((ql, qr, qm, r - 1, qk, xa, xb, nv))
        sol.append((ql * a + qr * b + qm * a % r * b + qk) % r)
    ccs = frontend.ConstraintSystem(r, ["p%d" % i for i in range(nb_public)], ["s0", "s1"], cons, "gates", len(sol))
    w = frontend.Witness(r, sol[:nb_public], sol[nb_public:nb_public + 2])
    return ccs, w, sol
