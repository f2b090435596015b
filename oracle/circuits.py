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
