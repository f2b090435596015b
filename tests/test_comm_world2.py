"""CPU tier: libapk's communicator (csrc/comm.cpp) in two and three PROCESSES without a GPU.  The C code under test is the
one a GPU node runs - TCP rendezvous and control plane, the dealing of a commitment batch by index range, the scatter of
the scalar slices (host-staged data plane: what two ranks sharing one GPU use as well), status words travelling with the
partial sums, the per-wire dealing, the worker loop - and only the GPU touch points are replaced through the `apk_compute`
seam of include/apk.h: "device" memory is host memory, the per-rank MSM and the coset NTT are the C oracle's."""
import ctypes as C
import multiprocessing as mp
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _compute_table(cv, clib, bases: bytes, n: int = 0, fail_msm: bool = False):
    """apk_compute over host memory: alloc / copy are malloc / memmove, msm_batch commits over `bases` (this rank's "context")
    with the C oracle, coset_ntt is the oracle's 4n coset NTT of the zero-padded polynomial."""
    from algoplonk_amd import _lib
    nb = 2 * cv.fp_bytes
    store = {}
    base_buf = C.create_string_buffer(bases, len(bases))

    def alloc(_u, nbytes, out):
        buf = C.create_string_buffer(max(nbytes, 1))
        store[C.addressof(buf)] = buf
        out[0] = C.addressof(buf)
        return 0

    def release(_u, p):
        store.pop(p, None)
        return 0

    def copy(_u, dst, src, nbytes, kind):
        C.memmove(dst, src, nbytes)
        return 0

    def msm(_u, basis, count, scalars, offsets, lens, out):
        if fail_msm:
            return _lib.APK_ERR_HIP
        for b in range(count):
            rc = clib.orc_msm(cv.abi, C.addressof(base_buf) + offsets[b] * nb, scalars[b], lens[b], 1, out + b * nb)
            if rc:
                return rc
        return 0

    def coset(_u, d_in, length, d_out):
        C.memset(d_out, 0, 4 * n * 32)
        C.memmove(d_out, d_in, length * 32)
        return clib.orc_ntt(cv.abi, d_out, 4 * n, 0, 1)

    cbs = (_lib.CP_MSM(msm), _lib.CP_COSET(coset), _lib.CP_ALLOC(alloc), _lib.CP_RELEASE(release), _lib.CP_COPY(copy))
    t = _lib.Compute()
    t.user, t.msm_batch, t.coset_ntt, t.alloc, t.release, t.copy, t.g1_bytes, t.n = None, cbs[0], cbs[1], cbs[2], cbs[3], cbs[4], nb, n
    return t, (cbs, store, base_buf)


def _setup_paths():
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))


def _sharded_worker(rank, world, port, q):
    _setup_paths()
    try:
        from algoplonk_amd import ecc, parallel
        from oracle import c_oracle, curves as oc, plonk as oplonk
        from oracle.prng import SplitMix64
        clib = c_oracle.load()
        comm = parallel.Comm(rank, world, "127.0.0.1", port)
        assert (comm.rank, comm.world) == (rank, world)
        for cv, ov in ((ecc.BN254, oc.BN254), (ecc.BLS12_381, oc.BLS12_381)):
            nb = 2 * cv.fp_bytes
            for n in (21, 2, 1):                    # 1 and 2 bases: ranks with an empty share send the point at infinity
                tau = 0x1234567
                g = SplitMix64(42)
                pts = cv.g1_vector([ov.mul(ov.g1, pow(tau, i, cv.r)) for i in range(n)])
                scalars = [g.fr(cv.r) for _ in range(n)]
                lo, hi = parallel.my_share(n, rank, world)
                table, keep = _compute_table(cv, clib, pts[lo * nb: hi * nb] or bytes(nb))
                comm.set_compute(table, keep)
                comm.bind(None)
                assert comm.transport == "tcp"
                mine = C.create_string_buffer(cv.fr_vector(scalars[lo:hi]) or b"\0")
                got = cv.g1_from_bytes(comm.msm_sharded(cv, C.addressof(mine), hi - lo))
                assert got == ov.mul(ov.g1, oplonk.poly_eval(scalars, tau, cv.r)), ("sharded MSM != full MSM", cv.name, n)
        # the timing protocol of bench.py: barrier, MAX over ranks
        comm.barrier()
        assert comm.max(float(rank)) == float(world - 1)
        comm.close()
        q.put((rank, "ok"))
    except Exception as e:
        import traceback
        q.put((rank, "FAIL: %r %s" % (e, traceback.format_exc()[-800:])))


def _split_worker(rank, world, port, q):
    _setup_paths()
    try:
        from algoplonk_amd import _lib, ecc, parallel
        from algoplonk_amd._lib import lib
        from oracle import c_oracle, curves as oc, plonk as oplonk
        from oracle.prng import SplitMix64
        clib = c_oracle.load()
        cv, ov = ecc.BLS12_381, oc.BLS12_381
        n = 8
        tau, nbases = 0xABCDEF12345, n + 3
        pts = cv.g1_vector([ov.mul(ov.g1, pow(tau, i, cv.r)) for i in range(nbases)])
        g = SplitMix64(7)
        vectors = [[g.fr(cv.r) for _ in range(m)] for m in (10, 10, 10, 11, 5)]
        comm = parallel.Comm(rank, world, "127.0.0.1", port)
        table, keep = _compute_table(cv, clib, pts, n=n)       # every rank holds the whole SRS (the circuit context)
        comm.set_compute(table, keep)
        comm.bind(None)
        steps = 0
        if rank == 0:
            host = [C.create_string_buffer(cv.fr_vector(v)) for v in vectors]
            ptr = [C.addressof(b) for b in host]
            # a batch of three (round 1 / round 3 of the prover), a single commitment, a batch of two of unequal lengths
            for batch in ([0, 1, 2], [3], [4, 3]):
                lens = [len(vectors[i]) for i in batch]
                got = comm.commit(cv, 0, [ptr[i] for i in batch], lens)
                want = [ov.mul(ov.g1, oplonk.poly_eval(vectors[i], tau, cv.r)) for i in batch]
                assert [cv.g1_from_bytes(b) for b in got] == want, batch
                steps += 1
            # (the replicated prover's step runs on every rank: below, after the leader / worker steps)
            # per-wire dealing: the 4n-coset evaluations of three canonical polynomials, polynomial i on rank i mod world
            evs = [C.create_string_buffer(4 * n * 32) for _ in range(3)]
            comm.wires([ptr[0], ptr[1], ptr[2]], [10, 10, 10], [C.addressof(e) for e in evs])
            steps += 1
            w4 = ov.omega(4 * n)
            for i in range(3):
                got = cv.fr_vector_decode(evs[i].raw)
                want = [oplonk.poly_eval(vectors[i], ov.coset_shift * pow(w4, j, cv.r) % cv.r, cv.r) for j in range(4 * n)]
                assert got == want, ("wire", i)
            comm.split_end()
        else:
            assert comm.serve() == 4
        # the replicated prover (apk_comm_spmd_begin): EVERY rank holds the vectors and calls the step; nothing is scattered and the
        # same sums come back on all ranks
        host = [C.create_string_buffer(cv.fr_vector(v)) for v in vectors]
        for batch in ([0, 1, 2], [3], [4, 3]):
            got = comm.commit_local(cv, 0, [C.addressof(host[i]) for i in batch], [len(vectors[i]) for i in batch])
            assert [cv.g1_from_bytes(b) for b in got] == [ov.mul(ov.g1, oplonk.poly_eval(vectors[i], tau, cv.r)) for i in batch], ("local", batch)
        # the sub-coset split's one exchange: an in-place all-gather of "device" memory (host-staged through the star on this tier)
        buf2 = C.create_string_buffer(world * 4096)
        C.memmove(C.addressof(buf2) + rank * 4096, bytes([(rank * 31 + i) & 0xFF for i in range(4096)]), 4096)
        comm.allgather_device(C.addressof(buf2), 4096)
        assert buf2.raw == b"".join(bytes([(r * 31 + i) & 0xFF for i in range(4096)]) for r in range(world)), "allgather_device"
        # round 5: a schedule says where its time went and why its data plane is not RCCL - on every rank
        ph = comm.phase_ms(reset=True)
        assert ph["commit_rounds"] >= 3 and ph["msm_ms"] > 0 and ph["sums_exchange_ms"] > 0 and ph["gathers"] == 1, ph
        assert comm.phase_ms()["commit_rounds"] == 0
        assert comm.transport == "tcp" and "host compute table" in comm.transport_reason, comm.transport_reason
        probe = comm.link_probe(1 << 16, 1 << 16)                 # no device memory on this tier: nothing to time
        assert probe["ring_send_recv_gbps"] == 0.0 and probe["allgather_gbps_received"] == 0.0, probe
        comm.close()
        # a rank that fails its share fails the step on EVERY rank, with its rank named
        comm = parallel.Comm(rank, world, "127.0.0.1", port + 1)
        table, keep = _compute_table(cv, clib, pts, n=n, fail_msm=(rank == world - 1))
        comm.set_compute(table, keep)
        comm.bind(None)
        if rank == 0:
            buf = C.create_string_buffer(cv.fr_vector(vectors[0]))
            with pytest.raises(_lib.ApkError, match="rank %d failed its share" % (world - 1)):
                comm.commit(cv, 0, [C.addressof(buf)], [10])
            comm.split_end()
        else:
            comm.serve()
        comm.close()
        q.put((rank, "ok"))
    except Exception as e:
        import traceback
        q.put((rank, "FAIL: %r %s" % (e, traceback.format_exc()[-800:])))


def _wires4_idle_worker(rank, world, port, q):
    """Two findings of the round-3 review, on the CPU tier: (1) FOUR wires at world 2 deal wires 1 and 3 to the one worker - the
    schedule must answer the first before the second arrives; (2) a leader that is idle for longer than APK_COMM_TIMEOUT_S between
    two steps must find its worker still serving (the timeout guards a step, not the wait for one)."""
    _setup_paths()
    try:
        os.environ["APK_COMM_TIMEOUT_S"] = "1"
        import time
        from algoplonk_amd import ecc, parallel
        from oracle import c_oracle, curves as oc, plonk as oplonk
        from oracle.prng import SplitMix64
        clib = c_oracle.load()
        cv, ov = ecc.BN254, oc.BN254
        n = 8
        tau = 0x5151
        pts = cv.g1_vector([ov.mul(ov.g1, pow(tau, i, cv.r)) for i in range(n + 3)])
        g = SplitMix64(11)
        vectors = [[g.fr(cv.r) for _ in range(m)] for m in (10, 9, 11, 10)]
        comm = parallel.Comm(rank, world, "127.0.0.1", port)
        table, keep = _compute_table(cv, clib, pts, n=n)
        comm.set_compute(table, keep)
        comm.bind(None)
        if rank == 0:
            host = [C.create_string_buffer(cv.fr_vector(v)) for v in vectors]
            ptr = [C.addressof(b) for b in host]
            w4 = ov.omega(4 * n)
            for rep in range(2):
                evs = [C.create_string_buffer(4 * n * 32) for _ in range(4)]
                comm.wires(ptr, [len(v) for v in vectors], [C.addressof(e) for e in evs])
                for i in range(4):
                    want = [oplonk.poly_eval(vectors[i], ov.coset_shift * pow(w4, j, cv.r) % cv.r, cv.r) for j in range(4 * n)]
                    assert cv.fr_vector_decode(evs[i].raw) == want, ("wire", i, rep)
                if rep == 0:
                    time.sleep(2.5)              # idle for 2.5 x the step timeout
            got = comm.commit(cv, 0, [ptr[0]], [10])
            assert cv.g1_from_bytes(got[0]) == ov.mul(ov.g1, oplonk.poly_eval(vectors[0], tau, cv.r))
            comm.split_end()
        else:
            assert comm.serve() == 3
        comm.close()
        q.put((rank, "ok"))
    except Exception as e:
        import traceback
        q.put((rank, "FAIL: %r %s" % (e, traceback.format_exc()[-800:])))


def _token_worker(rank, world, port, q):
    """A hello that does not carry the launch's APK_COMM_TOKEN is refused by rank 0."""
    _setup_paths()
    try:
        os.environ["APK_COMM_TOKEN"] = "launch-A" if rank == 0 else "launch-B"
        os.environ["APK_COMM_TIMEOUT_S"] = "5"
        from algoplonk_amd import _lib, parallel
        with pytest.raises(_lib.ApkError):
            c = parallel.Comm(rank, world, "127.0.0.1", port)
            c.barrier()                          # the refused worker finds its connection closed here at the latest
        q.put((rank, "ok"))
    except Exception as e:
        import traceback
        q.put((rank, "FAIL: %r %s" % (e, traceback.format_exc()[-800:])))


def _run(target, world, extra_ports=1):
    from algoplonk_amd.parallel import free_port
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    res = [q.get(timeout=240) for _ in procs]
    [p.join(timeout=60) for p in procs]
    assert sorted(res) == [(r, "ok") for r in range(world)], res


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_msm_over_the_c_communicator(world):
    """BASELINE.json configs[3]: ONE MSM split by index range, one all-gather of a point per rank, local additions.
    World 8 = the node the driver's scaling run uses: eight processes exist here for the first time, not on the hardware."""
    _run(_sharded_worker, world)


@pytest.mark.parametrize("world", [2, 3, 8])
def test_split_proof_schedule_over_the_c_communicator(world):
    """commitment batches dealt by index range + wires dealt by polynomial; leader / worker loop; the
    replicated prover's local commitments and the sub-coset split's all-gather (prove-split and prove-spmd of bench.py).  At
    world 8 five ranks get no wire and, in the batches of five to eleven scalars, some ranks an empty share."""
    _run(_split_worker, world)


def test_four_wires_at_world_2_and_an_idle_leader():
    _run(_wires4_idle_worker, 2)


def test_hello_without_the_launch_token_is_refused():
    _run(_token_worker, 2)


def test_from_env_rendezvous_file_is_private(tmp_path, monkeypatch):
    """Comm.from_env publishes port + token in a 0600 file inside a 0700 directory of the user; world 1 needs none."""
    sys.path.insert(0, ROOT)
    from algoplonk_amd import parallel
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setenv("RANK", "0")
    c = parallel.Comm.from_env()
    assert c.world == 1 and c.rccl_ranks == 0
    c.close()


def test_world_1_needs_no_peer():
    sys.path.insert(0, ROOT)
    from algoplonk_amd import parallel
    c = parallel.Comm(0, 1)
    c.barrier()
    assert c.max(3.5) == 3.5 and c.transport == "tcp"
    c.close()


def test_my_share_and_deal_tile_the_work():
    from algoplonk_amd.parallel import deal, my_share
    for total in (0, 1, 7, 8, 131072):
        for world in (1, 2, 3, 8):
            spans = [my_share(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
    for lens in ([18, 18, 18], [7], [5, 19], [1, 1, 1, 1]):
        segs = [deal(lens, r, 3) for r in range(3)]
        seen = sorted((k, i) for ss in segs for k, lo, hi in ss for i in range(lo, hi))
        assert seen == [(k, i) for k, n in enumerate(lens) for i in range(n)]
        sizes = [sum(hi - lo for _, lo, hi in ss) for ss in segs]
        assert max(sizes) - min(sizes) <= 1
