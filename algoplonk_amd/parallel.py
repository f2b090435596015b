"""Multi-GPU layout of the path: one process per GPU, libapk's OWN communicator (csrc/comm.cpp, declared in
include/apk.h "multi-GPU behind the boundary").  This module is a thin ctypes binding: every exchange step - the TCP control
plane, the RCCL data plane, the dealing and the error propagation - runs inside the library, so a cgo host
(/root/reference/algoplonk.go:89 is Go) calls exactly the same entry points.  No torch, no second HIP runtime.

* independent proofs             -> `my_share`: a contiguous share per rank, no data-path collective (replicas only).
* ONE MSM sharded by index range -> `ShardedMsm` (apk_msm_g1_sharded): rank r holds the windowed tables of bases [lo_r, hi_r)
  in its own HBM, commits its slice, ONE all-gather of a 64/96-byte point per rank, local EC additions (a numeric
  all-reduce cannot add curve points).  BASELINE.json configs[3].
* ONE proof on several GPUs      -> `Comm.split_begin / split_end / serve`: the commitment batches of the prover are dealt to
  the ranks by index range of their flattened (scalar, point) pairs; with APK_SPLIT_WIRES=1 the 4n-coset evaluations of the
  wire polynomials are dealt by wire (polynomial i -> rank i mod world).
"""
from __future__ import annotations

import ctypes as C
import os
import socket
import time
from typing import List, Optional, Sequence, Tuple

from . import ecc, _lib
from ._lib import lib, check


def my_share(total: int, rank: int, world: int) -> Tuple[int, int]:
    """[lo, hi) of `total` units owned by `rank`: sizes differ by at most one, earlier ranks take the remainder
    (the same rule as csrc/comm.cpp my_share)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def deal(lens: Sequence[int], rank: int, world: int):
    """The (commitment, lo, hi) segments of rank `rank` when the sum(lens) (scalar, point) pairs of a commitment batch are
    cut into `world` contiguous shares of the flattened pair list - the Python statement of csrc/comm.cpp deal(), kept for
    the tests that check the C schedule against it."""
    lo, hi = my_share(sum(lens), rank, world)
    segs, base = [], 0
    for k, n in enumerate(lens):
        a, b = max(lo, base), min(hi, base + n)
        if a < b:
            segs.append((k, a - base, b - base))
        base += n
    return segs


def g1_sum(curve: ecc.ID, points: bytes) -> bytes:
    """Sum of len(points) / (64|96) gnark-layout affine points in ONE library call (apk_g1_sum, host)."""
    nb = 2 * curve.fp_bytes
    out = C.create_string_buffer(nb)
    check(lib.apk_g1_sum(curve.abi, points, len(points) // nb, out))
    return out.raw


def g1_add(curve: ecc.ID, P: bytes, Q: bytes) -> bytes:
    """P + Q on gnark-layout affine points, using the library's own (host-instantiated) curve templates."""
    out = C.create_string_buffer(2 * curve.fp_bytes)
    check(lib.apk_host_g1_op(curve.abi, 0, P, Q, out))
    return out.raw


def rccl_selftest(device: int = 0) -> int:
    """World-1 RCCL communicator on `device` through every RCCL call of the data plane (apk_comm_rccl_selftest); returns
    ncclCommCount (1) or raises."""
    n = C.c_int32(0)
    check(lib.apk_comm_rccl_selftest(device, C.byref(n)))
    return n.value


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class Comm:
    """One rank of libapk's communicator (include/apk.h apk_comm_*)."""

    def __init__(self, rank: int, world: int, addr: str = "127.0.0.1", port: int = 0):
        self.rank, self.world = rank, world
        self._c = C.c_void_p()
        self._keep = []            # ctypes callbacks of a compute table must outlive the communicator
        check(lib.apk_comm_create(rank, world, addr.encode(), port, C.byref(self._c)))

    @classmethod
    def from_env(cls, tag: str = "") -> "Comm":
        """RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT as torchrun (or any launcher) exports them.  MASTER_PORT itself belongs
        to the launcher's own store, so rank 0 picks a free port and publishes it in a rendezvous file named after MASTER_PORT
        (one node: the contract of bench.py is N GPUs of ONE node)."""
        rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
        if world == 1:
            return cls(0, 1)
        addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
        # the ranks of one launch share their parent (the launcher's agent): its pid keeps a stale file of an earlier, crashed
        # launch on the same MASTER_PORT from being read
        # The file carries the port AND a random token of this launch (APK_COMM_TOKEN: libapk sends its hash in every hello and
        # rank 0 refuses connections without it).  It lives in a directory of its own (mode 0700, owner checked) and is created
        # with O_EXCL | O_NOFOLLOW, so a planted symlink or a pre-created file is an error, never followed.
        rdir = "/tmp/apk_rdzv_%s" % os.getuid()
        try:
            os.mkdir(rdir, 0o700)
        except FileExistsError:
            pass
        st = os.lstat(rdir)
        import stat as _stat
        if not _stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
            raise RuntimeError("rendezvous directory %s is not a private directory of this user" % rdir)
        path = os.path.join(rdir, "%s_%s%s" % (os.environ.get("MASTER_PORT", "0"), os.getppid(), tag))
        if rank == 0:
            token = os.environ.get("APK_COMM_TOKEN") or os.urandom(16).hex()
            os.environ["APK_COMM_TOKEN"] = token
            # rank 0 binds the port itself: picked free, then bound by apk_comm_create a moment later - a race with another
            # process taking it in between fails the create loudly (no silent cross-talk: the token guards the hello)
            port = free_port()
            for stale in (path, path + ".tmp"):      # left by a launch with the same parent pid that crashed during its rendezvous
                try:
                    os.unlink(stale)
                except OSError:
                    pass
            fd = os.open(path + ".tmp", os.O_WRONLY | os.O_CREAT | os.O_EXCL | os.O_NOFOLLOW, 0o600)
            with os.fdopen(fd, "w") as f:
                f.write("%d %s %d" % (port, token, os.getpid()))     # the pid lets a worker tell a live rank 0 from a stale file
            os.replace(path + ".tmp", path)
            try:
                return cls(0, world, addr, port)
            finally:
                try:
                    os.unlink(path)
                except OSError:
                    pass
        deadline = time.time() + 300
        while True:
            # a worker may get here before rank 0 has removed the file of an earlier, crashed launch: a file whose writer is no
            # longer alive is ignored and read again (rank 0 replaces it atomically)
            try:
                fd = os.open(path, os.O_RDONLY | os.O_NOFOLLOW)
                with os.fdopen(fd) as f:
                    port_s, token, pid_s = f.read().split()
                os.kill(int(pid_s), 0)
                break
            except (FileNotFoundError, ProcessLookupError, ValueError):
                pass
            except PermissionError:       # alive, another user's process: cannot be this launch's rank 0
                pass
            if time.time() > deadline:
                raise RuntimeError("rank %d: no rendezvous file %s from a live rank 0" % (rank, path))
            time.sleep(0.02)
        os.environ["APK_COMM_TOKEN"] = token
        return cls(rank, world, addr, int(port_s))

    def bind(self, ctx) -> "Comm":
        check(lib.apk_comm_bind(self._c, ctx))
        return self

    @property
    def transport(self) -> str:
        return lib.apk_comm_transport(self._c).decode()

    @property
    def rccl_ranks(self) -> int:
        """ncclCommCount of the data plane's RCCL communicator (0 unless the transport is "rccl")."""
        return lib.apk_comm_rccl_ranks(self._c)

    @property
    def transport_reason(self) -> str:
        """Why the data plane is not RCCL ("" when it is): apk_comm_transport_reason."""
        return lib.apk_comm_transport_reason(self._c).decode()

    def link_probe(self, ring_bytes: int = 64 << 20, gather_bytes: int = 32 << 20) -> dict:
        """One timed ring of ncclSend / ncclRecv and one all-gather over the data plane as it came up (apk_comm_link_probe,
        collective): what the first run on a multi-GPU node reads its link rate from."""
        ring, gather = C.c_double(0), C.c_double(0)
        check(lib.apk_comm_link_probe(self._c, ring_bytes, gather_bytes, C.byref(ring), C.byref(gather)))
        return {"ring_send_recv_gbps": round(ring.value, 2), "ring_bytes": ring_bytes, "allgather_gbps_received": round(gather.value, 2),
                "allgather_bytes_per_rank": gather_bytes}

    def phase_ms(self, reset: bool = False) -> dict:
        out = (C.c_double * 5)()
        check(lib.apk_comm_phase_ms(self._c, out, int(reset)))
        return {"msm_ms": out[0], "sums_exchange_ms": out[1], "subcoset_gather_ms": out[2], "commit_rounds": int(out[3]), "gathers": int(out[4])}

    def barrier(self) -> None:
        check(lib.apk_comm_barrier(self._c))

    def max(self, value: float) -> float:
        v = C.c_double(value)
        check(lib.apk_comm_max_f64(self._c, C.byref(v)))
        return v.value

    def set_compute(self, table: "_lib.Compute", keep=()) -> None:
        self._keep = [table] + list(keep)
        check(lib.apk_comm_set_compute(self._c, C.byref(table)))

    def msm_sharded(self, curve: ecc.ID, d_scalars, count: int) -> bytes:
        out = C.create_string_buffer(2 * curve.fp_bytes)
        check(lib.apk_msm_g1_sharded(self._c, d_scalars, count, out))
        return out.raw

    def split_begin(self) -> None:
        check(lib.apk_comm_split_begin(self._c))

    def split_end(self) -> None:
        check(lib.apk_comm_split_end(self._c))

    def spmd_begin(self) -> None:
        """Replicated prover: every rank calls this on its bound circuit context, then the same Prove; the commitments are
        shared out by index range with nothing scattered (apk_comm_spmd_begin)."""
        check(lib.apk_comm_spmd_begin(self._c))

    @property
    def subcoset_active(self) -> bool:
        return bool(lib.apk_comm_subcoset_active(self._c))

    def allgather_device(self, d_all: int, bytes_per_rank: int) -> None:
        """In-place all-gather of device memory (apk_comm_allgather_device): rank r's part at d_all + r * bytes_per_rank."""
        check(lib.apk_comm_allgather_device(self._c, d_all, bytes_per_rank))

    def spmd_end(self) -> None:
        check(lib.apk_comm_spmd_end(self._c))

    def commit_local(self, curve: ecc.ID, basis: int, d_scalars: Sequence[int], lens: Sequence[int]) -> List[bytes]:
        """The replicated prover's commitment step (apk_comm_commit_local): EVERY rank calls it with its own copy of the vectors."""
        k, nb = len(lens), 2 * curve.fp_bytes
        out = C.create_string_buffer(k * nb)
        check(lib.apk_comm_commit_local(self._c, basis, k, (C.c_void_p * k)(*d_scalars), (C.c_uint32 * k)(*lens), out))
        return [out.raw[i * nb:(i + 1) * nb] for i in range(k)]

    def serve(self) -> int:
        steps = C.c_uint64(0)
        check(lib.apk_comm_serve(self._c, C.byref(steps)))
        return steps.value

    def commit(self, curve: ecc.ID, basis: int, d_scalars: Sequence[int], lens: Sequence[int]) -> List[bytes]:
        """The leader's commitment step as the prover's hook calls it (apk_comm_commit)."""
        k, nb = len(lens), 2 * curve.fp_bytes
        ptrs = (C.c_void_p * k)(*d_scalars)
        ls = (C.c_uint32 * k)(*lens)
        out = C.create_string_buffer(k * nb)
        check(lib.apk_comm_commit(self._c, basis, k, ptrs, ls, out))
        return [out.raw[i * nb:(i + 1) * nb] for i in range(k)]

    def wires(self, d_canonical: Sequence[int], lens: Sequence[int], d_evals: Sequence[int]) -> None:
        k = len(lens)
        check(lib.apk_comm_wires(self._c, k, (C.c_void_p * k)(*d_canonical), (C.c_uint32 * k)(*lens), (C.c_void_p * k)(*d_evals)))

    def close(self) -> None:
        if self._c:
            lib.apk_comm_destroy(self._c)
            self._c = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ShardedMsm:
    """sum_i s_i * P_i with the index range split across the ranks of a communicator: rank r keeps the windowed tables of
    bases [lo_r, hi_r) resident in its own HBM (an MSM-only libapk context) and its slice of the scalars in device memory."""

    def __init__(self, curve: ecc.ID, bases: bytes, device: int, comm: Comm, msm_window: int = 0, share: Optional[Tuple[int, int]] = None):
        """`share` = (rank, world) overrides the communicator's for the slicing only: single-process tests hold several
        "ranks" on one GPU with world-1 communicators and add the partial sums themselves."""
        self.curve, self.comm = curve, comm
        nb = 2 * curve.fp_bytes
        self.total = len(bases) // nb
        self.lo, self.hi = my_share(self.total, *(share or (comm.rank, comm.world)))
        self._ctx = C.c_void_p()
        self._d = C.c_void_p()
        # an empty share still binds a (one-base) context: the rank takes part in the exchange with the point at infinity
        lo, hi = (self.lo, self.hi) if self.hi > self.lo else (0, 1)
        check(lib.apk_msm_ctx_create(curve.abi, device, bases[lo * nb: hi * nb], hi - lo, msm_window, C.byref(self._ctx)))
        comm.bind(self._ctx)

    def upload(self, scalars: bytes) -> None:
        """Place this rank's slice of the FULL Montgomery scalar vector in device memory (inputs resident in HBM)."""
        mine = scalars[self.lo * 32: self.hi * 32]
        if self._d:
            check(lib.apk_device_free(self._ctx, self._d))
            self._d = C.c_void_p()
        if mine:
            check(lib.apk_device_alloc(self._ctx, len(mine), C.byref(self._d)))
            check(lib.apk_device_upload(self._ctx, self._d, mine, len(mine)))

    def run(self) -> bytes:
        """One sharded MSM over the uploaded slices: the same affine point on every rank."""
        return self.comm.msm_sharded(self.curve, self._d, self.hi - self.lo)

    def close(self) -> None:
        if self._ctx:
            self.comm.bind(None)                  # the communicator's staging buffers were allocated through this context
            if self._d:
                lib.apk_device_free(self._ctx, self._d)
            lib.apk_ctx_destroy(self._ctx)
            self._ctx = C.c_void_p()
