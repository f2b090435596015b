"""Multi-GPU layout of the path: one process per GPU over torch.distributed (backend "nccl" = RCCL on
ROCm; "gloo" in the CPU tests).

* independent proofs            -> `my_share`: a contiguous share per rank, no data-path collective (replicas only).
* ONE MSM sharded by index range -> `ShardedMsm`: rank r holds the windowed tables for bases [lo_r, hi_r) in its own
  HBM, computes a full partial MSM over its slice of the scalars, then ONE all-gather of world x (64|96)-byte affine
  points and world-1 point additions give the result on every rank.  A numeric all-reduce cannot add curve points,
  hence all-gather of bytes + local EC addition; the payload is O(100 B), so xGMI bandwidth is irrelevant and only the
  ~10 us collective latency counts (BASELINE.json configs[3]).
"""
from __future__ import annotations

import ctypes as C
from typing import Tuple

from . import ecc
from ._lib import lib, check


def my_share(total: int, rank: int, world: int) -> Tuple[int, int]:
    """[lo, hi) of `total` units owned by `rank`: sizes differ by at most one, earlier ranks take the remainder."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def g1_add(curve: ecc.ID, P: bytes, Q: bytes) -> bytes:
    """P + Q on gnark-layout affine points, using the library's own (host-instantiated) curve templates."""
    out = C.create_string_buffer(2 * curve.fp_bytes)
    check(lib.apk_host_g1_op(curve.abi, 0, P, Q, out))
    return out.raw


def g1_sum(curve: ecc.ID, points: bytes) -> bytes:
    """Sum of len(points) / (64|96) gnark-layout affine points in ONE library call (apk_g1_sum, host)."""
    nb = 2 * curve.fp_bytes
    out = C.create_string_buffer(nb)
    check(lib.apk_g1_sum(curve.abi, points, len(points) // nb, out))
    return out.raw


def gather_and_add(curve: ecc.ID, partial: bytes, group=None) -> bytes:
    """The one exchange step of a sharded MSM: all-gather one affine point per rank into ONE (world x 64|96)-byte tensor,
    one device-to-host copy, one library call for the world-1 point additions (same result on every rank)."""
    import torch
    import torch.distributed as dist

    nb = 2 * curve.fp_bytes
    world = dist.get_world_size(group)
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    mine = torch.frombuffer(bytearray(partial), dtype=torch.uint8).to(dev)
    allpts = torch.empty(world * nb, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(allpts, mine, group=group)
    return g1_sum(curve, allpts.cpu().numpy().tobytes())


class ShardedMsm:
    """sum_i s_i * P_i with the index range split across the ranks of a process group: rank r keeps the windowed tables of
    bases [lo_r, hi_r) resident in its own HBM (an MSM-only libapk context)."""

    def __init__(self, curve: ecc.ID, bases: bytes, device: int, rank: int, world: int, msm_window: int = 0):
        self.curve, self.rank, self.world = curve, rank, world
        nb = 2 * curve.fp_bytes
        self.total = len(bases) // nb
        self.lo, self.hi = my_share(self.total, rank, world)
        self._ctx = C.c_void_p()
        if self.hi > self.lo:
            self._open(bases[self.lo * nb: self.hi * nb], device, msm_window)

    def _open(self, my_bases: bytes, device: int, msm_window: int) -> None:
        check(lib.apk_msm_ctx_create(self.curve.abi, device, my_bases, self.hi - self.lo, msm_window, C.byref(self._ctx)))

    def local_partial(self, mine: bytes) -> bytes:
        """Partial sum over this rank's `mine` = scalars[lo:hi] (Montgomery bytes), on this rank's GPU."""
        out = C.create_string_buffer(2 * self.curve.fp_bytes)
        check(lib.apk_msm_g1(self._ctx, 0, mine, self.hi - self.lo, out))
        return out.raw

    def partial(self, scalars: bytes) -> bytes:
        """This rank's partial sum over its index range; `scalars` = the FULL Montgomery scalar vector."""
        if self.hi == self.lo:
            return bytes(2 * self.curve.fp_bytes)
        return self.local_partial(scalars[self.lo * 32: self.hi * 32])

    def run(self, scalars: bytes, group=None) -> bytes:
        return gather_and_add(self.curve, self.partial(scalars), group)

    def close(self) -> None:
        if self._ctx:
            lib.apk_ctx_destroy(self._ctx)
            self._ctx = C.c_void_p()
