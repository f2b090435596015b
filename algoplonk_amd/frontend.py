"""Minimal stand-in for the part of gnark's frontend the reference drives:
`frontend.Compile(field, scs.NewBuilder, circuit)` (/root/reference/algoplonk.go:50) and
`frontend.NewWitness(assignment, field)` (/root/reference/algoplonk.go:81).

In the Go integration this stays gnark (OUT OF SCOPE for the GPU path): the shim hands
libapk the trace columns and the solved L/R/O vectors gnark already holds.  With no Go toolchain in this image
the Python mirror needs *something* that produces those same arrays, so this module implements just enough of
a PLONK (SparseR1CS) builder to express the reference's test circuits:

    class Pythagorean(Circuit):                     # examples/basic/logicsigVerifier/main.go:27-43
        A = Public(); B = Public(); C = Secret()
        def define(self, api):
            api.AssertIsEqual(api.Add(api.Mul(self.A, self.A), api.Mul(self.B, self.B)), api.Mul(self.C, self.C))

Gate i:  ql*a + qr*b + qm*a*b + qo*c + qk == 0.  Variables are numbered public, secret, internal (gnark SCS
wire order [UPSTREAM]); public rows are `ql = -1` placeholders; the permutation is gnark's trace.S.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple

from . import ecc


class Public:
    """Declares a public input of a circuit (gnark struct tag `gnark:",public"`)."""


class Secret:
    """Declares a secret input."""


@dataclass(frozen=True)
class Variable:
    wire: int


class Circuit:
    """Subclass, declare inputs as class attributes (Public()/Secret()), implement define(api)."""

    def define(self, api: "API") -> None:  # pragma: no cover - interface
        raise NotImplementedError

    @classmethod
    def _inputs(cls) -> Tuple[List[str], List[str]]:
        pub, sec = [], []
        for klass in reversed(cls.__mro__):
            for k, v in vars(klass).items():
                if isinstance(v, Public):
                    pub.append(k)
                elif isinstance(v, Secret):
                    sec.append(k)
        return pub, sec


@dataclass
class ConstraintSystem:
    """What `frontend.Compile` returns: cs.SparseR1CS restricted to what the prover path consumes."""
    field: int
    public_names: List[str]
    secret_names: List[str]
    constraints: List[Tuple[int, int, int, int, int, int, int, int]]  # ql,qr,qm,qo,qk, xa,xb,xc
    solver: List[Tuple[int, Callable[[List[int]], int]]]              # (wire, value from earlier wires)
    nb_variables: int
    # BSB22 (frontend.Committer): per commitment (rows of the committed constraints, row of the commitment constraint)
    # = gnark constraint.PlonkCommitment{Committed, CommitmentIndex} [UPSTREAM]
    commitments: List[Tuple[List[int], int]] = field(default_factory=list)

    def GetNbPublicVariables(self) -> int:
        return len(self.public_names)

    def GetNbConstraints(self) -> int:
        return len(self.constraints)

    def domain_size(self) -> int:
        """ecc.NextPowerOfTwo(nbConstraints + nbPublic) (/root/reference/setup/setup.go:113-114), floor 8."""
        m = self.GetNbConstraints() + self.GetNbPublicVariables()
        n = 1
        while n < m:
            n *= 2
        return max(n, 8)


class API:
    def __init__(self, field_mod: int, nb_inputs: int):
        self.r = field_mod
        self.next_wire = nb_inputs
        self.constraints: List[Tuple[int, int, int, int, int, int, int, int]] = []
        self.solver: List[Tuple[int, Callable[[List[int]], int]]] = []
        self.commitments: List[Tuple[List[int], int]] = []

    def _new(self, fn: Callable[[List[int]], int]) -> Variable:
        w = self.next_wire
        self.next_wire += 1
        self.solver.append((w, fn))
        return Variable(w)

    def Mul(self, a: Variable, b: Variable) -> Variable:
        r = self.r
        c = self._new(lambda s, a=a.wire, b=b.wire: s[a] * s[b] % r)
        self.constraints.append((0, 0, 1, r - 1, 0, a.wire, b.wire, c.wire))
        return c

    def Add(self, a: Variable, b: Variable) -> Variable:
        r = self.r
        c = self._new(lambda s, a=a.wire, b=b.wire: (s[a] + s[b]) % r)
        self.constraints.append((1, 1, 0, r - 1, 0, a.wire, b.wire, c.wire))
        return c

    def Sub(self, a: Variable, b: Variable) -> Variable:
        r = self.r
        c = self._new(lambda s, a=a.wire, b=b.wire: (s[a] - s[b]) % r)
        self.constraints.append((1, r - 1, 0, r - 1, 0, a.wire, b.wire, c.wire))
        return c

    def Gate(self, ql: int, qr: int, qm: int, qk: int, a: Variable, b: Variable) -> Variable:
        """c = ql*a + qr*b + qm*a*b + qk (one custom gate; used by the random benchmark circuits)."""
        r = self.r
        c = self._new(lambda s, a=a.wire, b=b.wire: (ql * s[a] + qr * s[b] + qm * s[a] % r * s[b] + qk) % r)
        self.constraints.append((ql % r, qr % r, qm % r, r - 1, qk % r, a.wire, b.wire, c.wire))
        return c

    def AssertIsEqual(self, a: Variable, b: Variable) -> None:
        self.constraints.append((1, self.r - 1, 0, 0, 0, a.wire, b.wire, 0))

    def AssertIsDifferent(self, a: Variable, zero: int = 0) -> None:
        """Only the `AssertIsDifferent(x, 0)` form the reference's circuits use (bsb22_test.go:34): x * inv == 1."""
        if zero != 0:
            raise NotImplementedError("AssertIsDifferent is implemented against the constant 0 only")
        r = self.r
        inv = self._new(lambda s, a=a.wire: pow(s[a], -1, r))
        self.constraints.append((0, 0, 1, 0, r - 1, a.wire, inv.wire, 0))

    def Commit(self, *vs: Variable) -> Variable:
        """frontend.Committer.Commit (bsb22_test.go:30): gnark's scs builder emits, per committed variable, the
        constraint  -v + qcp*pi2 = 0  and then  -cmt + qk = 0  whose qk the prover and the verifier inject as
        hash_fr([pi2])  [UPSTREAM frontend/cs/scs]."""
        r = self.r
        rows = []
        for v in vs:
            rows.append(len(self.constraints))
            self.constraints.append((r - 1, 0, 0, 0, 0, v.wire, 0, 0))
        k = len(self.commitments)
        cmt = self.next_wire
        self.next_wire += 1
        self.solver.append((cmt, ("commit", k)))
        self.commitments.append((rows, len(self.constraints)))
        self.constraints.append((r - 1, 0, 0, 0, 0, cmt, 0, 0))
        return Variable(cmt)


def Compile(field_mod: int, circuit: Circuit) -> ConstraintSystem:
    """frontend.Compile(curve.ScalarField(), scs.NewBuilder, circuit) (/root/reference/algoplonk.go:50)."""
    pub, sec = circuit._inputs()
    api = API(field_mod, len(pub) + len(sec))
    shadow = type(circuit).__new__(type(circuit))
    shadow.__dict__.update(circuit.__dict__)
    for i, name in enumerate(pub + sec):
        setattr(shadow, name, Variable(i))
    shadow.define(api)
    return ConstraintSystem(field_mod, pub, sec, api.constraints, api.solver, api.next_wire, api.commitments)


@dataclass
class Witness:
    """frontend.NewWitness(assignment, field): the input vector, public first (algoplonk.go:81-85)."""
    field: int
    public: List[int]
    secret: List[int]

    def Public(self) -> "Witness":
        return Witness(self.field, list(self.public), [])

    def Vector(self) -> List[int]:
        return self.public + self.secret


def NewWitness(assignment: Circuit, field_mod: int) -> Witness:
    pub, sec = assignment._inputs()
    vals = []
    for name in pub + sec:
        v = getattr(assignment, name)
        if isinstance(v, (Public, Secret)):
            raise ValueError("error creating witness: %s is not assigned" % name)
        vals.append(int(v) % field_mod)
    return Witness(field_mod, vals[: len(pub)], vals[len(pub):])


def solve(ccs: ConstraintSystem, w: Witness, commit_hint=None, hiding=None, pi2_out=None) -> List[int]:
    """Full variable assignment (gnark's constraint solver, SURVEY.md §3.3 R1 `solveConstraints`).
    BSB22: `commit_hint(column) -> Fr` is gnark's hint (commit the column over the Lagrange SRS, hash the point with
    hash_fr); `hiding[k]` = the two random entries gnark places in the column; the columns are appended to pi2_out."""
    s = w.Vector() + [0] * (ccs.nb_variables - len(w.public) - len(w.secret))
    if ccs.solver == "gates":
        # every constraint defines its own output wire: c = ql*a + qr*b + qm*a*b + qk  (qo = -1)
        r = ccs.field
        for ql, qr, qm, qo, qk, xa, xb, xc in ccs.constraints:
            s[xc] = (ql * s[xa] + qr * s[xb] + qm * s[xa] % r * s[xb] + qk) % r
        return s
    n, nbp = ccs.domain_size(), ccs.GetNbPublicVariables()
    for wire, fn in ccs.solver:
        if isinstance(fn, tuple) and fn[0] == "commit":
            if commit_hint is None:
                raise ValueError("circuit uses Commit: a commitment hint is required to solve it")
            k = fn[1]
            rows, cidx = ccs.commitments[k]
            col = [0] * n
            for row in rows:
                col[nbp + row] = s[ccs.constraints[row][5]]
            col[nbp + cidx] = hiding[k][0]
            col[nbp + ccs.GetNbConstraints() - 1] = hiding[k][1]
            s[wire] = commit_hint(col)
            if pi2_out is not None:
                pi2_out.append(col)
        elif isinstance(fn, tuple) and fn[0] == "gate":
            # closure-free encoding used by the large synthetic workloads: c = ql*a + qr*b + qm*a*b + qk
            _, ql, qr, qm, qk, xa, xb = fn
            s[wire] = (ql * s[xa] + qr * s[xb] + qm * s[xa] % ccs.field * s[xb] + qk) % ccs.field
        elif isinstance(fn, tuple) and fn[0] == "inv":
            s[wire] = pow(s[fn[1]], -1, ccs.field)
        else:
            s[wire] = fn(s)
    return s


# ---- trace + wire columns: the arrays that cross the C-ABI (include/apk.h apk_circuit_desc / apk_prove) ----

@dataclass
class Trace:
    n: int
    ql: List[int]
    qr: List[int]
    qm: List[int]
    qo: List[int]
    qk: List[int]
    perm: List[int]
    qcp: List[List[int]] = field(default_factory=list)


def build_trace(ccs: ConstraintSystem) -> Trace:
    """gnark plonk.NewTrace + buildPermutation [UPSTREAM]; see SURVEY.md §3.1."""
    r = ccs.field
    n = ccs.domain_size()
    nbp = ccs.GetNbPublicVariables()
    ql, qr, qm, qo, qk = ([0] * n for _ in range(5))
    lro = [0] * (3 * n)
    for i in range(nbp):
        ql[i] = r - 1
        lro[i] = i
    for j, (a, b, m, o, k, xa, xb, xc) in enumerate(ccs.constraints):
        i = nbp + j
        ql[i], qr[i], qm[i], qo[i], qk[i] = a, b, m, o, k
        lro[i], lro[n + i], lro[2 * n + i] = xa, xb, xc
    perm = [-1] * (3 * n)
    last = [-1] * max(ccs.nb_variables, 1)
    for i, v in enumerate(lro):
        if last[v] != -1:
            perm[i] = last[v]
        last[v] = i
    for i, v in enumerate(lro):
        if perm[i] == -1:
            perm[i] = last[v]
    qcp = []
    for rows, _ in ccs.commitments:
        col = [0] * n
        for row in rows:
            col[nbp + row] = 1
        qcp.append(col)
    return Trace(n, ql, qr, qm, qo, qk, perm, qcp)


def wire_columns(ccs: ConstraintSystem, solution: Sequence[int]) -> Tuple[List[int], List[int], List[int]]:
    """L, R, O in Lagrange form; placeholder/padding cells carry variable 0 (gnark evaluateLROSmallDomain)."""
    n = ccs.domain_size()
    nbp = ccs.GetNbPublicVariables()
    s0 = solution[0] if solution else 0
    L, R, O = [s0] * n, [s0] * n, [s0] * n
    for i in range(nbp):
        L[i] = solution[i]
    for j, c in enumerate(ccs.constraints):
        i = nbp + j
        L[i], R[i], O[i] = solution[c[5]], solution[c[6]], solution[c[7]]
    return L, R, O
