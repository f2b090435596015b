"""Multi-GPU layout of the path: one process per GPU over torch.distributed (backend "nccl" = RCCL on
ROCm; "gloo" in the CPU tests).

* independent proofs            -> `my_share`: a contiguous share per rank, no data-path collective (replicas only).
* ONE proof on several GPUs      -> `SplitCommitter`: every commitment batch of the prover is dealt to the ranks by index
  range of its flattened (scalar, point) pairs; the transcript and the polynomial arithmetic stay on rank 0 (SURVEY.md §8e
  row 2: "independent commitments inside one proof").
* ONE MSM sharded by index range -> `ShardedMsm`: rank r holds the windowed tables for bases [lo_r, hi_r) in its own
  HBM, computes a full partial MSM over its slice of the scalars, then ONE all-gather of world x (64|96)-byte affine
  points and world-1 point additions give the result on every rank.  A numeric all-reduce cannot add curve points,
  hence all-gather of bytes + local EC addition; the payload is O(100 B), so xGMI bandwidth is irrelevant and only the
  ~10 us collective latency counts (BASELINE.json configs[3]).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

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


# ---- intra-proof multi-GPU: the commitments of ONE proof dealt to the ranks -----------------------------
def deal(lens: Sequence[int], rank: int, world: int):
    """The (commitment, lo, hi) segments of rank `rank` when the sum(lens) (scalar, point) pairs of a commitment batch are
    cut into `world` contiguous shares of the flattened pair list (sizes differ by at most one)."""
    lo, hi = my_share(sum(lens), rank, world)
    segs, base = [], 0
    for k, n in enumerate(lens):
        a, b = max(lo, base), min(hi, base + n)
        if a < b:
            segs.append((k, a - base, b - base))
        base += n
    return segs


class _DeviceBytes:
    """`n` bytes of this context's GPU memory with the one method SplitCommitter needs (data_ptr): the single-rank path
    stays free of torch - its wheel bundles a ROCm runtime of its own, and the second HIP runtime to open the GPU in one
    process does not see it, so torch may only be brought in where RCCL needs it (world > 1; bench.py initialises it first)."""

    def __init__(self, ctx, n: int):
        self.ctx, self.p = ctx, C.c_void_p()
        check(lib.apk_device_alloc(ctx, max(n, 16), C.byref(self.p)))

    def data_ptr(self) -> int:
        return self.p.value

    def __del__(self):
        try:
            if self.p:
                lib.apk_device_free(self.ctx, self.p)
        except Exception:
            pass


class SplitCommitter:
    """One proof, several GPUs.  Every rank holds the circuit context (SRS tables resident: any rank can commit any index
    range of any polynomial); rank 0 runs the prover (transcript, NTTs, quotient ...) and at each Fiat-Shamir sync point its
    commit hook (include/apk.h apk_ctx_set_commit_hook) deals the batch's (scalar, point) pairs evenly:

        leader : header broadcast (basis, lens)  ->  scatter of the scalar slices (one RCCL scatter: the slices leave rank 0
                 on its 7 xGMI links in parallel; 2^21 x 32 B x 3 polynomials / 8 ranks = 24 MB per link)
        all    : apk_msm_g1_batch_device over the dealt segments (<= 3 segments per rank)
        all    : all-gather of count x (64|96)-byte partial sums; the leader adds them per commitment (apk_g1_sum)

    Workers sit in serve() until the leader sends the stop header.  The only collectives are that broadcast, scatter and
    all-gather; a numeric all-reduce cannot add curve points."""

    STOP = -1

    def __init__(self, curve: ecc.ID, ctx, rank: int, world: int, group=None, device: Optional[str] = None):
        self.curve, self.ctx, self.rank, self.world, self.group = curve, ctx, rank, world, group
        self.torch = self.dist = None
        self.dev = device
        if world > 1:
            import torch
            import torch.distributed as dist
            self.torch, self.dist = torch, dist
            # staging tensors live where the collectives run: the GPU (RCCL) - "cpu" only for the gloo tier tests, which replace
            # both GPU touch points (fill_chunks / local_commit)
            self.dev = device or ("cuda" if dist.get_backend(group) == "nccl" else "cpu")
            if self.dev == "cuda" and not torch.cuda.is_available():
                raise RuntimeError("SplitCommitter: torch does not see the GPU - import torch and touch torch.cuda BEFORE libapk's "
                                   "first HIP call (two HIP runtimes in one process: the second one finds no device)")
        self._hook = None
        self.batches = 0

    # -- what a rank does with its share: overridden in the CPU tests (no GPU there)
    def local_commit(self, basis: int, chunk, segs, lens) -> bytes:
        """Partial sums of this rank's segments: chunk = its scalar slices back to back (a uint8 tensor on this rank's GPU).
        Returns len(lens) affine points, infinity where the rank holds no part of a commitment."""
        cv = self.curve
        nb = 2 * cv.fp_bytes
        out = bytearray(len(lens) * nb)
        if segs:
            k = len(segs)
            ptrs, offs, ls, at = (C.c_void_p * k)(), (C.c_uint64 * k)(), (C.c_uint64 * k)(), 0
            for i, (_, lo, hi) in enumerate(segs):
                ptrs[i], offs[i], ls[i] = chunk.data_ptr() + at, lo, hi - lo
                at += (hi - lo) * 32
            res = C.create_string_buffer(k * nb)
            check(lib.apk_msm_g1_batch_device(self.ctx, basis, k, ptrs, offs, ls, res))
            for i, (c, _, _) in enumerate(segs):
                out[c * nb: (c + 1) * nb] = res.raw[i * nb: (i + 1) * nb]
        return bytes(out)

    def fill_chunks(self, staging, d_scalars, lens, chunk_bytes: int) -> None:
        """Leader: copy every rank's slices out of the prover's vectors into that rank's row of the staging tensor."""
        for r in range(self.world):
            at = r * chunk_bytes
            for k, lo, hi in deal(lens, r, self.world):
                check(lib.apk_device_copy(self.ctx, staging.data_ptr() + at, d_scalars[k] + lo * 32, (hi - lo) * 32))
                at += (hi - lo) * 32

    def _round(self, basis: int, lens, d_scalars=None) -> Optional[List[bytes]]:
        torch, dist, cv = self.torch, self.dist, self.curve
        nb = 2 * cv.fp_bytes
        total = sum(lens)
        chunk_bytes = ((total + self.world - 1) // self.world) * 32
        if self.world == 1:                        # no collective: the whole batch is this rank's share
            staging = _DeviceBytes(self.ctx, chunk_bytes)
            self.fill_chunks(staging, d_scalars, lens, chunk_bytes)
            part = self.local_commit(basis, staging, deal(lens, 0, 1), lens)
            self.batches += 1
            return [g1_sum(cv, part[c * nb: (c + 1) * nb]) for c in range(len(lens))]
        mine = torch.empty(chunk_bytes, dtype=torch.uint8, device=self.dev)
        if self.rank == 0:
            staging = torch.empty(self.world * chunk_bytes, dtype=torch.uint8, device=self.dev)
            self.fill_chunks(staging, d_scalars, lens, chunk_bytes)
            dist.scatter(mine, [staging[r * chunk_bytes: (r + 1) * chunk_bytes] for r in range(self.world)], src=0, group=self.group)
        else:
            dist.scatter(mine, None, src=0, group=self.group)
        part = self.local_commit(basis, mine, deal(lens, self.rank, self.world), lens)
        allp = torch.empty(self.world * len(lens) * nb, dtype=torch.uint8, device=self.dev)
        dist.all_gather_into_tensor(allp, torch.frombuffer(bytearray(part), dtype=torch.uint8).to(self.dev), group=self.group)
        raw = allp.cpu().numpy().tobytes()
        self.batches += 1
        if self.rank != 0:
            return None
        k = len(lens)
        return [g1_sum(cv, b"".join(raw[(r * k + c) * nb: (r * k + c + 1) * nb] for r in range(self.world))) for c in range(k)]

    def _header(self, basis: int, lens) -> List[int]:
        if self.world == 1:
            return [basis, len(lens)] + list(lens)
        torch, dist = self.torch, self.dist
        h = torch.zeros(8, dtype=torch.int64, device=self.dev)
        if self.rank == 0:
            h[0], h[1] = basis, len(lens)
            for i, n in enumerate(lens):
                h[2 + i] = n
        dist.broadcast(h, src=0, group=self.group)
        return [int(x) for x in h.cpu().tolist()]

    # -- leader
    def commit(self, basis: int, d_scalars: Sequence[int], lens: Sequence[int]) -> List[bytes]:
        self._header(basis, lens)
        return self._round(basis, list(lens), list(d_scalars))

    def install(self) -> None:
        """Leader: route the context's commitments through this object (apk_ctx_set_commit_hook)."""
        from . import _lib
        nb = 2 * self.curve.fp_bytes

        def hook(_user, basis, count, d_scalars, lens, out_points):
            try:
                pts = self.commit(basis, [d_scalars[i] for i in range(count)], [lens[i] for i in range(count)])
                C.memmove(out_points, b"".join(pts), count * nb)
                return 0
            except Exception as e:  # never unwind through the C frames
                import sys
                print("SplitCommitter hook: %r" % (e,), file=sys.stderr)
                return 3
        self._hook = _lib.COMMIT_HOOK(hook)
        check(lib.apk_ctx_set_commit_hook(self.ctx, self._hook, None))

    def stop(self) -> None:
        if self.rank == 0:
            if self._hook is not None:
                check(lib.apk_ctx_set_commit_hook(self.ctx, type(self._hook)(0), None))
                self._hook = None
            if self.world > 1:
                self._header(self.STOP, [])

    # -- workers
    def serve(self) -> int:
        """Rank != 0: take part in the leader's rounds until it stops; returns the number of batches served."""
        while True:
            h = self._header(0, [])
            if h[0] == self.STOP:
                return self.batches
            self._round(h[0], h[2: 2 + h[1]])
