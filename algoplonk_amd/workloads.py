"""Synthetic workloads of BASELINE.md §2 (seeded, no network, no ceremony files): the circuits and witnesses that
bench.py and the full-size tests prove.  Host-side only - produces the arrays gnark's frontend + solver would hand
to the prover (trace columns, permutation, solved L/R/O, public inputs, blinding scalars).
"""
from __future__ import annotations

import hashlib
from dataclasses import dataclass
from typing import List, Tuple

from . import ecc, frontend

MASK64 = (1 << 64) - 1


class SplitMix64:
    """Same generator as the oracle's ("SplitMix64 -> Fr by rejection")."""

    def __init__(self, seed: int):
        self.s = seed & MASK64

    def next(self) -> int:
        self.s = (self.s + 0x9E3779B97F4A7C15) & MASK64
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
        return z ^ (z >> 31)

    def fr(self, r: int) -> int:
        bits = r.bit_length()
        while True:
            v = 0
            for i in range(4):
                v |= self.next() << (64 * i)
            v &= (1 << bits) - 1
            if v < r:
                return v

    def below(self, n: int) -> int:
        lim = (1 << 64) - ((1 << 64) % n)
        while True:
            v = self.next()
            if v < lim:
                return v % n


def tau_from_seed(seed: int, r: int) -> int:
    return int.from_bytes(hashlib.sha256(seed.to_bytes(8, "big")).digest(), "big") % r


@dataclass
class Workload:
    name: str
    curve: ecc.ID
    ccs: frontend.ConstraintSystem
    witness: frontend.Witness
    solution: List[int]
    blinding: List[int]
    tau: int


@dataclass
class Variant:
    """One assignment of a circuit: the inputs, the solved variables, the 9 blinding scalars (and BSB22 hiding pairs)."""
    witness: frontend.Witness
    blinding: List[int]
    solution: List[int] = None
    hiding: List[Tuple[int, int]] = None


def variant_inputs(ccs: frontend.ConstraintSystem, count: int, seed: int) -> List[Variant]:
    """`count` assignments of ONE circuit (same selectors, same permutation - one proving context), every one with its own
    input vector, its own blinding and its own BSB22 hiding pairs, all from SplitMix64(seed + i): what concurrent callers of
    (*CompiledCircuit).Verify hand the prover (/root/reference/algoplonk.go:79-98 - each call has its own assignment).  The
    synthetic circuits take ANY inputs (every gate defines its output wire), so each variant is a satisfying witness; unsolved
    here (BSB22 circuits need the context's commitment hint: plonk.solve_with_commitments)."""
    r = ccs.field
    nbp, nbs = ccs.GetNbPublicVariables(), len(ccs.secret_names)
    out = []
    for i in range(count):
        g = SplitMix64((seed + 0x9E37 * (i + 1)) & MASK64)
        vals = [g.fr(r) for _ in range(nbp + nbs)]
        gb = SplitMix64(((seed + 0x9E37 * (i + 1)) ^ 0xB11D) & MASK64)
        blinding = [gb.fr(r) for _ in range(9)]
        hiding = [(gb.fr(r), gb.fr(r)) for _ in range(len(ccs.commitments))]
        out.append(Variant(frontend.Witness(r, vals[:nbp], vals[nbp:]), blinding, None, hiding))
    return out


def variants(wl: "Workload", count: int, seed: int) -> List[Variant]:
    """The workload's own assignment followed by count - 1 others of the same circuit (variant_inputs), solved."""
    out = [Variant(wl.witness, wl.blinding, wl.solution, [])]
    for v in variant_inputs(wl.ccs, count - 1, seed):
        v.solution = frontend.solve(wl.ccs, v.witness)
        out.append(v)
    return out


def random_circuit_bsb22(curve: ecc.ID, log_n: int, seed: int, nb_commitments: int = 1, committed: int = 16,
                         nb_public: int = 2):
    """BASELINE.json configs[4] shape: the random-gate circuit with `nb_commitments` BSB22 commitments, each over
    `committed` wires (the committed column is all zero except those entries + 2 hiding values), each commitment feeding
    later gates.  Returns (ccs, witness, blinding, tau); solving needs the commitment hint (plonk.Prove supplies it)."""
    r = curve.r
    n = 1 << log_n
    g = SplitMix64(seed)
    inputs = [g.fr(r) for _ in range(nb_public + 2)]
    cons, solver, commitments = [], [], []
    nv = nb_public + 2
    budget = n - nb_public
    per_commit = committed + 2                      # committed rows + the commitment row + the != 0 row
    plain = budget - nb_commitments * per_commit
    assert plain > 4 * nb_commitments
    next_commit_at = plain // (nb_commitments + 1)
    done_commits = 0
    for i in range(plain):
        xa, xb = g.below(nv), g.below(nv)
        ql, qr, qm, qk = g.fr(r), g.fr(r), g.fr(r), g.fr(r)
        cons.append((ql, qr, qm, r - 1, qk, xa, xb, nv))
        solver.append((nv, ("gate", ql, qr, qm, qk, xa, xb)))
        nv += 1
        if done_commits < nb_commitments and i + 1 == next_commit_at * (done_commits + 1):
            rows = []
            for _ in range(committed):
                rows.append(len(cons))
                cons.append((r - 1, 0, 0, 0, 0, g.below(nv), 0, 0))       # -v + qcp*pi2 = 0
            cmt = nv
            solver.append((cmt, ("commit", done_commits)))
            commitments.append((rows, len(cons)))
            cons.append((r - 1, 0, 0, 0, 0, cmt, 0, 0))                   # -cmt + qk(injected) = 0
            inv = nv + 1
            solver.append((inv, ("inv", cmt)))
            cons.append((0, 0, 1, 0, r - 1, cmt, inv, 0))                 # cmt * inv - 1 = 0
            nv += 2
            done_commits += 1
    ccs = frontend.ConstraintSystem(r, ["p%d" % i for i in range(nb_public)], ["s0", "s1"], cons, solver, nv, commitments)
    w = frontend.Witness(r, inputs[:nb_public], inputs[nb_public:])
    gb = SplitMix64(seed ^ 0xB11D)
    return ccs, w, [gb.fr(r) for _ in range(9)], tau_from_seed(seed, r)


def skewed_circuit(curve: ecc.ID, log_n: int, seed: int, nb_public: int = 2) -> Workload:
    """The hard input of SURVEY.md section 7 ("Lagrange-basis MSMs take witness values: many zeros / ones / small values"): a
    circuit shaped like bit-decomposition-heavy gadgets (range checks, SHA-style logic) - half of the gates are XOR / AND over
    boolean wires, a quarter multiplex 16-bit values by a boolean, a quarter are the uniform random gates of `random_circuit`.
    Solved wire columns: ~80 % in {0, 1} (AND gates drift towards 0), the rest 16-bit or uniform values.  (The reference's own MiMC-Merkle example,
    examples/merkle/logicsigVerifier/main.go:45-61, is the opposite extreme: 17 levels x 2 MiMC permutations of field-sized
    values and 16 path bits - more than 99 % of its wires are uniform.)"""
    r = curve.r
    n = 1 << log_n
    g = SplitMix64(seed)
    sol = [g.fr(r) for _ in range(nb_public + 2)]
    cons = []
    m1 = r - 1
    bools, smalls = [], []

    def emit(ql, qr, qm, qk, xa, xb, pool=None):
        nv = len(sol)
        a, b = sol[xa], sol[xb]
        cons.append((ql % r, qr % r, qm % r, m1, qk % r, xa, xb, nv))
        sol.append((ql * a + qr * b + qm * a % r * b + qk) % r)
        if pool is not None:
            pool.append(nv)

    budget = n - nb_public
    for i in range(8):                                  # constant wires: the seeds of the boolean and the 16-bit pools
        emit(0, 0, 0, i & 1, 0, 0, bools)
        emit(0, 0, 0, g.below(1 << 16), 0, 0, smalls)
    while len(cons) < budget:
        kind = g.below(4)
        if kind < 2:
            xa, xb = bools[g.below(len(bools))], bools[g.below(len(bools))]
            if g.below(2):
                emit(1, 1, -2, 0, xa, xb, bools)        # XOR
            else:
                emit(0, 0, 1, 0, xa, xb, bools)         # AND
        elif kind == 2:
            emit(0, 0, 1, 0, smalls[g.below(len(smalls))], bools[g.below(len(bools))], smalls)   # 16-bit value AND-ed with a bit
        else:
            nv = len(sol)
            emit(g.fr(r), g.fr(r), g.fr(r), g.fr(r), g.below(nv), g.below(nv))
    ccs = frontend.ConstraintSystem(r, ["p%d" % i for i in range(nb_public)], ["s0", "s1"], cons, "gates", len(sol))
    w = frontend.Witness(r, sol[:nb_public], sol[nb_public:nb_public + 2])
    gb = SplitMix64(seed ^ 0xB11D)
    return Workload("%s bit-heavy circuit (~80%% of the wire values in {0,1}), 2^%d constraints" % (curve.name, log_n), curve, ccs, w,
                    sol, [gb.fr(r) for _ in range(9)], tau_from_seed(seed, r))


def random_circuit(curve: ecc.ID, log_n: int, seed: int, nb_public: int = 2) -> Workload:
    """BASELINE.json configs[1]/[2]: n - nb_public random gates c = ql*a + qr*b + qm*a*b + qk over earlier wires."""
    r = curve.r
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
    gb = SplitMix64(seed ^ 0xB11D)
    return Workload("%s random circuit, 2^%d constraints" % (curve.name, log_n), curve, ccs, w, sol,
                    [gb.fr(r) for _ in range(9)], tau_from_seed(seed, r))
