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


def _split_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from algoplonk_amd import ecc, parallel
    from oracle import curves as oc, plonk as oplonk
    from oracle.prng import SplitMix64

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cv, ov = ecc.BLS12_381, oc.BLS12_381
        tau, nbases = 0xABCDEF12345, 19
        pts = [ov.mul(ov.g1, pow(tau, i, cv.r)) for i in range(nbases)]
        g = SplitMix64(7)
        vectors = [[g.fr(cv.r) for _ in range(m)] for m in (18, 18, 18, 19, 5)]
        host = [cv.fr_vector(v) for v in vectors]

        class OracleSplit(parallel.SplitCommitter):
            """The product's scheduling (deal / header / scatter / all-gather / per-commitment sums) with the two GPU
            touch points swapped: slices are cut from host bytes, the per-rank partial MSM is the oracle's."""
            def fill_chunks(self, staging, d_scalars, lens, chunk_bytes):
                for r in range(self.world):
                    at = r * chunk_bytes
                    for k, lo, hi in parallel.deal(lens, r, self.world):
                        staging[at: at + (hi - lo) * 32] = torch.frombuffer(bytearray(host[d_scalars[k]][lo * 32: hi * 32]), dtype=torch.uint8)
                        at += (hi - lo) * 32

            def local_commit(self, basis, chunk, segs, lens):
                raw, at, out = chunk.numpy().tobytes(), 0, [None] * len(lens)
                for k, lo, hi in segs:
                    sc = cv.fr_vector_decode(raw[at: at + (hi - lo) * 32])
                    out[k] = ov.msm_naive(pts[lo:hi], sc)
                    at += (hi - lo) * 32
                return b"".join(cv.g1_to_bytes(P) for P in out)

        sc = OracleSplit(cv, None, rank, world, device="cpu")
        if rank == 0:
            # a batch of three (round 1 / round 3 of the prover), a single commitment, a batch of two of unequal lengths
            for batch in ([0, 1, 2], [3], [4, 3]):
                lens = [len(vectors[i]) for i in batch]
                got = sc.commit(0, batch, lens)
                want = [ov.mul(ov.g1, oplonk.poly_eval(vectors[i], tau, cv.r)) for i in batch]
                assert [cv.g1_from_bytes(b) for b in got] == want, batch
            sc.stop()
            assert sc.batches == 3
        else:
            assert sc.serve() == 3
        # the deal tiles the flattened pair list: every pair exactly once, shares differ by at most one pair
        for lens in ([18, 18, 18], [7], [5, 19], [1, 1, 1, 1]):
            segs = [parallel.deal(lens, r, 3) for r in range(3)]
            seen = sorted((k, i) for ss in segs for k, lo, hi in ss for i in range(lo, hi))
            assert seen == [(k, i) for k, n in enumerate(lens) for i in range(n)]
            sizes = [sum(hi - lo for _, lo, hi in ss) for ss in segs]
            assert max(sizes) - min(sizes) <= 1
        q.put((rank, "ok"))
    except Exception as e:
        import traceback
        q.put((rank, "FAIL: %r %s" % (e, traceback.format_exc()[-600:])))
    finally:
        dist.destroy_process_group()


def test_split_committer_schedule_world2():
    """SURVEY.md section 8e row 2 (intra-proof multi-GPU): leader / worker protocol of SplitCommitter on two gloo ranks."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_split_worker, args=(r, 2, port, q)) for r in range(2)]
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
