"""CPU tier: the N>1 layout on world_size-2 gloo.  The collective step (all-gather of per-rank partial points + local
EC additions through the library's host-instantiated curve code) is the real one; only the per-rank partial MSM -
a GPU kernel - is replaced by the oracle so the test runs without a GPU."""
import os
import socket
import sys

import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from algoplonk_amd import ecc, parallel
    from oracle import curves as oc, plonk as oplonk
    from oracle.prng import SplitMix64

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        for cv, ov in ((ecc.BN254, oc.BN254), (ecc.BLS12_381, oc.BLS12_381)):
            n, tau = 21, 0x1234567
            g = SplitMix64(42)
            pts = [ov.mul(ov.g1, pow(tau, i, cv.r)) for i in range(n)]
            scalars = [g.fr(cv.r) for _ in range(n)]

            class OracleShard(parallel.ShardedMsm):
                """The product class with its per-rank GPU MSM swapped for the oracle's (no GPU in this tier)."""
                def _open(self, my_bases, device, msm_window):
                    self._pts = cv.g1_vector_decode(my_bases)

                def local_partial(self, mine, cv=cv, ov=ov):
                    return cv.g1_to_bytes(ov.msm_naive(self._pts, cv.fr_vector_decode(mine)))

            sm = OracleShard(cv, cv.g1_vector(pts), device=0, rank=rank, world=world)
            assert (sm.lo, sm.hi) == parallel.my_share(n, rank, world)
            got = cv.g1_from_bytes(sm.run(cv.fr_vector(scalars)))
            assert got == ov.mul(ov.g1, oplonk.poly_eval(scalars, tau, cv.r)), "sharded MSM != full MSM"
            # degenerate: a rank with an empty share contributes the point at infinity
            sm2 = OracleShard(cv, cv.g1_vector(pts[:1]), device=0, rank=rank, world=world)
            assert cv.g1_from_bytes(sm2.run(cv.fr_vector(scalars[:1]))) == ov.mul(pts[0], scalars[0])
        # proof sharding: shares tile [0, total) without gaps
        lo, hi = parallel.my_share(11, rank, world)
        import torch
        t = torch.tensor([hi - lo])
        dist.all_reduce(t)
        assert int(t.item()) == 11
        q.put((rank, "ok"))
    except Exception as e:  # surface the failure in the parent
        q.put((rank, "FAIL: %r" % (e,)))
    finally:
        dist.destroy_process_group()


def test_sharded_msm_and_shares_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = [q.get(timeout=240) for _ in procs]
    [p.join(timeout=60) for p in procs]
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def test_my_share_partitions():
    from algoplonk_amd.parallel import my_share
    for total in (0, 1, 7, 8, 131072):
        for world in (1, 2, 3, 8):
            spans = [my_share(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
