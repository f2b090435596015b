"""ORACLE (test infrastructure): the test circuits of the reference, restated as SparseR1CS rows.

* pythagorean   examples/basic/logicsigVerifier/main.go:27-43, assignment (3,4,5) at :49-52
* identity      compile_test.go:13-20
* square        bsb22_test.go:18-39 without the Commit calls (X == Y*Y)
* random_chain  BASELINE.md §2 "BN254 random circuit" (seeded; every gate's output is a fresh
                variable, inputs are earlier variables so the permutation has real cycles)
Variables are numbered public first, then secret, then internal - gnark's SCS wire order [UPSTREAM].
"""
from __future__ import annotations

from typing import List, Tuple

from .curves import Curve
from .plonk import Circuit
from .prng import SplitMix64


def pythagorean(cv: Curve, a=3, b=4, c=5) -> Tuple[Circuit, List[int]]:
    m1 = cv.r - 1
    # vars: 0=A 1=B (public) 2=C 3=aa 4=bb 5=cc 6=aa+bb
    cons = [
        (0, 0, 1, m1, 0, 0, 0, 3),
        (0, 0, 1, m1, 0, 1, 1, 4),
        (0, 0, 1, m1, 0, 2, 2, 5),
        (1, 1, 0, m1, 0, 3, 4, 6),
        (1, m1, 0, 0, 0, 6, 5, 0),
    ]
    sol = [a, b, c, a * a % cv.r, b * b % cv.r, c * c % cv.r, (a * a + b * b) % cv.r]
    return Circuit(cv, 2, 7, cons), sol


def identity(cv: Curve, x=7) -> Tuple[Circuit, List[int]]:
    return Circuit(cv, 1, 1, [(1, cv.r - 1, 0, 0, 0, 0, 0, 0)]), [x % cv.r]


def square(cv: Curve, x=9, y=3) -> Tuple[Circuit, List[int]]:
    # X (public, var 0) == Y*Y (var 1)
    return Circuit(cv, 1, 2, [(0, 0, 1, cv.r - 1, 0, 1, 1, 0)]), [x % cv.r, y % cv.r]


def random_chain(cv: Curve, log_n: int, seed: int, nb_public: int = 2) -> Tuple[Circuit, List[int]]:
    r = cv.r
    n = 1 << log_n
    m = n - nb_public
    g = SplitMix64(seed)
    sol = [g.fr(r) for _ in range(nb_public + 2)]  # publics + two secret seeds
    cons = []
    for _ in range(m):
        nv = len(sol)
        xa, xb = g.below(nv), g.below(nv)
        ql, qr, qm, qk = g.fr(r), g.fr(r), g.fr(r), g.fr(r)
        a, b = sol[xa], sol[xb]
        c = (ql * a + qr * b + qm * a % r * b + qk) % r
        cons.append((ql, qr, qm, r - 1, qk, xa, xb, nv))
        sol.append(c)
    return Circuit(cv, nb_public, len(sol), cons), sol


def bsb22_square(cv: Curve, nb_commitments: int, x: int = 9, y: int = 3):
    """bsb22_test.go:18-39: X == Y*Y, then `Commit(Y, X)` nb_commitments times, each followed by AssertIsDifferent(cmt, 0).
    Constraint encoding of gnark's scs builder `Commit` [UPSTREAM frontend/cs/scs, restated; SURVEY.md §3.3 R1]:
      committed wire v :  -v + qcp*pi2 = 0          (ql = -1, qcp row = 1)
      commitment wire  :  -cmt + qk = 0             (ql = -1, qk injected by prover and verifier = hash_fr([pi2]))
    Returns (circuit, partial solution with the commitment-dependent wires unset, plan) - see `solve_bsb22`."""
    from .plonk import Commitment
    r = cv.r
    m1 = r - 1
    # vars: 0 = X (public), 1 = Y, then per commitment: cmt_k, inv_k
    cons = [(0, 0, 1, m1, 0, 1, 1, 0)]                      # Y*Y - X = 0
    commitments = []
    plan = []
    nv = 2
    for _ in range(nb_commitments):
        committed_rows = []
        for v in (1, 0):                                    # Commit(Y, X)
            committed_rows.append(len(cons))
            cons.append((m1, 0, 0, 0, 0, v, 0, 0))
        cmt, inv = nv, nv + 1
        nv += 2
        cidx = len(cons)
        cons.append((m1, 0, 0, 0, 0, cmt, 0, 0))            # -cmt + qk(injected) = 0
        cons.append((0, 0, 1, 0, m1, cmt, inv, 0))          # cmt * inv - 1 = 0   (AssertIsDifferent(cmt, 0))
        commitments.append(Commitment(committed_rows, cidx))
        plan.append((cmt, inv))
    return Circuit(cv, 1, nv, cons, commitments), [x % r, y % r] + [0] * (nv - 2), plan


def solve_bsb22(c: Circuit, sol, plan, commit_lagrange, hiding):
    """gnark's solver with the BSB22 hint: for each commitment build the committed column pi2
    (wire values at the committed rows + two hiding entries), commit it over the Lagrange SRS, hash the point to Fr
    (hash_fr) and assign that to the commitment wire.  `commit_lagrange(evals) -> point`; hiding = 2 scalars each."""
    from .plonk import hash_fr
    cv = c.curve
    r = cv.r
    n = c.domain_size()
    off = c.nb_public
    sol = list(sol)
    pi2s = []
    for k, cm in enumerate(c.commitments):
        col = [0] * n
        for row in cm.committed:
            col[off + row] = sol[c.constraints[row][5]]
        col[off + cm.commitment_index] = hiding[k][0]
        col[off + len(c.constraints) - 1] = hiding[k][1]
        P = commit_lagrange(col)
        val = hash_fr(cv.raw_bytes(P), r)
        cmt, inv = plan[k]
        sol[cmt] = val
        sol[inv] = pow(val, -1, r)
        pi2s.append(col)
    return sol, pi2s
