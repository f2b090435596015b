"""Several assignments of ONE circuit prepared for concurrent provers: what a proving service holds when many
`(*CompiledCircuit).Verify` calls (/root/reference/algoplonk.go:79-98 - each with its own assignment) arrive for the same
compiled circuit.  Host-side packing only: every byte of arithmetic stays behind the C-ABI.

A `WitnessSet` keeps, per assignment, the solved wire columns L, R, O, the public inputs, the blinding scalars and the BSB22
columns in gnark's in-memory layout, in up to three places:

  * ordinary (pageable) host memory       - `apk_prove` as a cgo host calls it with plain Go slices;
  * page-locked host memory               - `apk_prove` with buffers from `apk_host_alloc` (the shim's witness pool);
  * device memory of the context's GPU    - `apk_prove_device` (inputs resident in HBM: bench.py's `value`).

bench.py, tools/soak.py and the under-load parity tests all draw their callers' inputs from here, so that no two concurrent
callers share a witness unless the caller asks for it.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

from . import _lib, frontend, plonk
from ._lib import lib, check


@dataclass
class Assignment:
    L: bytes
    R: bytes
    O: bytes
    public: bytes
    blinding: bytes
    pi2: List[bytes] = field(default_factory=list)
    dev: Optional[list] = None          # device pointers [L, R, O], then one per BSB22 column
    pinned: Optional[list] = None       # page-locked host pointers, same order
    public_ints: List[int] = field(default_factory=list)


class WitnessSet:
    def __init__(self, pk: Optional["plonk.ProvingKey"], ccs: frontend.ConstraintSystem, variants: Sequence, curve=None):
        """variants: workloads.Variant list (witness, blinding, solution or None, hiding).  Unsolved variants are solved here;
        circuits with BSB22 commitments go through the context's commitment hint (plonk.solve_with_commitments).
        pk = None (with `curve`): packing only, for circuits without commitments - nothing can be placed or proved."""
        self.pk, self.ccs, self.cv = pk, ccs, (pk.curve if pk is not None else curve)
        cv = self.cv
        self.items: List[Assignment] = []
        for v in variants:
            pi2_cols: List[List[int]] = []
            if ccs.commitments:
                solution, pi2_cols = plonk.solve_with_commitments(ccs, pk, v.witness, v.hiding)
            else:
                solution = v.solution if v.solution is not None else frontend.solve(ccs, v.witness)
            L, R, O = frontend.wire_columns(ccs, solution)
            self.items.append(Assignment(cv.fr_vector(L), cv.fr_vector(R), cv.fr_vector(O), cv.fr_vector(v.witness.public),
                                         cv.fr_vector(v.blinding), [cv.fr_vector(c) for c in pi2_cols],
                                         public_ints=list(v.witness.public)))

    def __len__(self) -> int:
        return len(self.items)

    # ---- placement ---------------------------------------------------------------------------------------------------
    def to_device(self) -> "WitnessSet":
        for it in self.items:
            if it.dev is None:
                it.dev = []
                for b in [it.L, it.R, it.O] + it.pi2:
                    p = C.c_void_p()
                    check(lib.apk_device_alloc(self.pk.ctx, len(b), C.byref(p)))
                    check(lib.apk_device_upload(self.pk.ctx, p, b, len(b)))
                    it.dev.append(p)
        return self

    def to_pinned(self, device: int = 0) -> "WitnessSet":
        for it in self.items:
            if it.pinned is None:
                it.pinned = []
                for b in [it.L, it.R, it.O] + it.pi2:
                    p = C.c_void_p()
                    check(lib.apk_host_alloc(device, len(b), C.byref(p)))
                    C.memmove(p, b, len(b))
                    it.pinned.append(p)
        return self

    def corrupt(self, index: int, row: Optional[int] = None) -> None:
        """Make assignment `index` UNSATISFYING: one output-wire value off by one in every copy of it (a gate no longer holds)."""
        it = self.items[index]
        n = len(it.O) // 32
        row = n // 2 if row is None else row
        o = bytearray(it.O)
        o[32 * row] ^= 1
        it.O = bytes(o)
        if it.dev is not None:
            check(lib.apk_device_upload(self.pk.ctx, it.dev[2], it.O, len(it.O)))
        if it.pinned is not None:
            C.memmove(it.pinned[2], it.O, len(it.O))

    # ---- one proof ---------------------------------------------------------------------------------------------------
    def prove(self, index: int, out: "_lib.Proof", where: str = "device") -> int:
        """apk_prove_device (where = "device") or apk_prove from page-locked ("pinned") / ordinary ("pageable") host memory.
        Returns the library's status code."""
        it = self.items[index]
        nb = len(it.pi2)
        if where == "device":
            ptrs = it.dev
            pi2 = (C.c_void_p * nb)(*ptrs[3:]) if nb else None
            return lib.apk_prove_device(self.pk.ctx, ptrs[0], ptrs[1], ptrs[2], it.public, it.blinding, pi2, C.byref(out))
        if where == "pinned":
            ptrs = it.pinned
            pi2 = (C.c_void_p * nb)(*ptrs[3:]) if nb else None
            return lib.apk_prove(self.pk.ctx, ptrs[0], ptrs[1], ptrs[2], it.public, it.blinding, pi2, C.byref(out))
        if where == "pageable":
            pi2 = (C.c_void_p * nb)(*[C.cast(C.c_char_p(b), C.c_void_p) for b in it.pi2]) if nb else None
            return lib.apk_prove(self.pk.ctx, it.L, it.R, it.O, it.public, it.blinding, pi2, C.byref(out))
        raise ValueError("where = device | pinned | pageable")

    def close(self) -> None:
        for it in self.items:
            if it.dev is not None:
                for p in it.dev:
                    lib.apk_device_free(self.pk.ctx, p)
                it.dev = None
            if it.pinned is not None:
                for p in it.pinned:
                    lib.apk_host_free(p)
                it.pinned = None
