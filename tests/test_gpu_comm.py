"""GPU tier: libapk's communicator with real contexts.  The GPU boxes of this tier have ONE MI355X, so two (and three) ranks
share device 0: RCCL refuses two ranks on one device, the communicator notices (apk_comm_bind compares the ranks' device
ordinals) and moves the data plane to HIP IPC pulls (or, IPC off / refused, stages it through the host and its TCP star) -
every other line of csrc/comm.cpp (the dealing,
apk_msm_g1_batch_device over dealt index ranges, the per-wire dealing through apk_coset_ntt_device, the worker loop, the commit
and wire hooks inside apk_prove) is what an 8-GPU node runs."""
import ctypes as C
import multiprocessing as mp
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q, cname, split_wires, ipc):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import faulthandler
    faulthandler.dump_traceback_later(200, exit=False, file=sys.stderr)     # a silent rank shows where it sits before the parent gives up
    stage = lambda s: q.put((rank, "@" + s))                                # progress markers: a timeout names the step it happened in
    try:
        os.environ["APK_COMM_IPC"] = "1" if ipc else "0"
        os.environ["APK_COMM_TIMEOUT_S"] = "30"          # a rank that dies must not keep its peers (and the GPU box) waiting
        if split_wires:
            os.environ["APK_SPLIT_WIRES"] = "1"
        from algoplonk_amd import MarshalProof, _lib, parallel, plonk as ap_plonk, setup as ap_setup
        from helpers import CURVES, blinding, random_chain_ccs
        from oracle import plonk as oplonk
        from oracle.prng import SplitMix64, tau_from_seed
        cv, ov = CURVES[cname]
        stage("imported")
        comm = parallel.Comm(rank, world, "127.0.0.1", port)
        stage("rendezvous")
        # ---- ONE proof, its commitments (and wires) dealt to the ranks: every rank holds the circuit context
        ccs, w, sol = random_chain_ccs(cv, 10, 0xA190 + 10)
        tau = tau_from_seed(99, cv.r)
        srs = ap_setup.unsafe_srs(cv, ccs.domain_size(), tau, device=0)
        pk, vk = ap_plonk.Setup(ccs, srs, device=0)
        stage("setup")
        comm.bind(pk.ctx)
        stage("bound " + comm.transport)
        # two ranks on one device: RCCL refuses; HIP IPC (the receiver maps the sender's staging buffer and pulls) unless it was
        # switched off or the box refuses to share the allocation - then the host-staged TCP star
        assert comm.transport in (("ipc", "tcp") if ipc else ("tcp",)), comm.transport
        transport = comm.transport
        bl = blinding(cv, 5)
        if rank == 0:
            plain = MarshalProof(ap_plonk.Prove(ccs, pk, w, bl))
            comm.split_begin()
            for _ in range(2):
                assert MarshalProof(ap_plonk.Prove(ccs, pk, w, bl)) == plain, "split proof differs from the single-GPU proof"
            comm.split_end()
            assert MarshalProof(ap_plonk.Prove(ccs, pk, w, bl)) == plain
        else:
            served = comm.serve()
            assert served == 2 * (4 + (1 if split_wires else 0)), served      # {L,R,O} {Z} {H1..3} {W, W'} (+ the wires) per proof
        stage("split proofs")
        comm.barrier()
        # ---- the replicated prover (apk_comm_spmd_begin): EVERY rank proves, commitments shared out by index range from each
        # rank's own polynomials, nothing scattered; every rank must return the single-GPU proof
        plain_here = MarshalProof(ap_plonk.Prove(ccs, pk, w, bl))
        comm.spmd_begin()
        assert comm.subcoset_active == (world in (2, 4, 8)), "round 3 on sub-cosets whenever the world is 2, 4 or 8"
        for _ in range(2):
            assert MarshalProof(ap_plonk.Prove(ccs, pk, w, bl)) == plain_here, "replicated-prover proof differs from the single-GPU proof"
        comm.spmd_end()
        assert MarshalProof(ap_plonk.Prove(ccs, pk, w, bl)) == plain_here
        stage("replicated prover")
        comm.barrier()
        # ---- ONE MSM sharded by index range (BASELINE.json configs[3]): MSM-only context over this rank's slice of the bases
        n = ccs.domain_size()
        g = SplitMix64(0xA192)
        scalars = [g.fr(cv.r) for _ in range(n + 3)]
        comm2 = parallel.Comm(rank, world, "127.0.0.1", port + 1)
        sm = parallel.ShardedMsm(cv, srs.g1, device=0, comm=comm2)
        sm.upload(cv.fr_vector(scalars))
        got = cv.g1_from_bytes(sm.run())
        assert got == ov.mul(ov.g1, oplonk.poly_eval(scalars, tau, cv.r)), "sharded MSM != full MSM"
        sm.close()
        comm2.close()
        comm.close()
        pk.close()
        faulthandler.cancel_dump_traceback_later()
        q.put((rank, "ok " + transport))
    except Exception as e:
        import traceback
        q.put((rank, "FAIL: %r %s" % (e, traceback.format_exc()[-900:])))


@pytest.mark.gpu
@pytest.mark.parametrize("cname,world,split_wires,ipc", [("bn254", 2, False, True), ("bls12-381", 2, True, True), ("bn254", 3, True, True),
                                                         ("bn254", 2, True, False), ("bls12-381", 4, False, True)])
def test_split_proof_and_sharded_msm_between_processes_sharing_the_gpu(gpu, cname, world, split_wires, ipc):
    from algoplonk_amd.parallel import free_port
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, cname, split_wires, ipc)) for r in range(world)]
    [p.start() for p in procs]
    import queue
    import time
    res, last, deadline = [], {}, time.monotonic() + 600           # a fresh box pages the image in under three processes at once
    while len(res) < world:
        try:
            r = q.get(timeout=max(1.0, deadline - time.monotonic()))
        except queue.Empty:
            [p.kill() for p in procs]
            pytest.fail("ranks silent for 600 s; finished %r, last stage per rank %r" % (res, last))
        if r[1].startswith("@"):
            last[r[0]] = r[1][1:]
        else:
            res.append(r)
    [p.join(timeout=60) for p in procs]
    assert all(r[1].startswith("ok") for r in res) and sorted(r[0] for r in res) == list(range(world)), res
    assert len({r[1] for r in res}) == 1, res        # every rank agreed on the data plane
    print("data plane:", res[0][1][3:])


@pytest.mark.gpu
@pytest.mark.parametrize("mode,extra,gpus", [("prove-split", ["--curve", "bls12_381", "--log-n", "12"], 2), ("msm-sharded", ["--log-n", "12"], 2),
                                             ("prove-spmd", ["--curve", "bls12_381", "--log-n", "12"], 2),
                                             ("prove", ["--log-n", "12", "--inflight", "2"], 2),
                                             # the world the driver's scaling run uses: EIGHT ranks exist here first (sharing the one GPU
                                             # over HIP IPC), so that the first real 8-GPU run exercises nothing but RCCL itself
                                             ("msm-sharded", ["--log-n", "12"], 8), ("prove-split", ["--log-n", "12"], 8),
                                             ("prove-spmd", ["--curve", "bls12_381", "--log-n", "12"], 8),
                                             ("prove", ["--log-n", "12", "--inflight", "2"], 8)])
def test_bench_py_runs_its_multi_rank_modes_under_the_contract_launcher(gpu, mode, extra, gpus):
    """`python bench.py --gpus N ...` re-executes itself under torch.distributed.run (the contract's command line), the N ranks
    rendezvous on libapk's communicator and rank 0 prints ONE JSON line with n_gpus = N.  Functional only: the ranks share the
    box's one GPU (APK_BENCH_SHARE_GPU=1), so the numbers mean nothing."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["APK_BENCH_SHARE_GPU"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--mode", mode, "--steps", "3", "--warmup", "1",
                        "--no-pmc", "--no-cpu-baseline", "--no-oracle-check", "--no-host-inputs"] + extra, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = lines[0]
    assert line["n_gpus"] == gpus and line["config"]["world_size"] == gpus and line["steps"] == 3
    if mode in ("prove-split", "prove-spmd"):
        assert line["matches_single_gpu_proof"] is True and line["scaling"] == "strong"
    if mode == "prove":
        assert line["scaling"] == "weak" and line["value"] > 0 and line["proofs_under_load_ok_on_all_ranks"] is True
    # round 5: every multi-rank line says what its data plane really was and measures it (two ranks on ONE GPU cannot be RCCL:
    # the line must say so, with the reason, instead of passing for an xGMI measurement)
    plane = line["config"]["data_plane"]
    assert plane["rccl_ranks"] == 0 and plane["transport"].startswith("fallback:ipc:") and "share device 0" in plane["transport"], plane
    assert plane["link_gbps"]["allgather_gbps_received"] > 0 and plane["link_gbps"]["ring_send_recv_gbps"] == 0.0, plane
    if mode == "prove-spmd":
        ph = line["phase_ms_per_proof_rank0"]
        assert ph["msm_ms"] > 0 and ph["sums_exchange_ms"] > 0 and ph["commit_rounds_per_proof"] == 4.0 and ph["non_msm_ms"] > 0, ph
        assert ph["subcoset_split"] is True and ph["subcoset_gather_ms"] > 0, ph        # round 3 on sub-cosets at world 2 and world 8


@pytest.mark.gpu
def test_bench_line_carries_the_contract_fields(gpu):
    """One small default-mode run of bench.py (2^12, a few steps): ONE JSON line with the contract's keys, the `roofline` object of
    the dominant kernel (HIP-event launch time, algorithmic bytes, PMC traffic when rocprofv3 is there) and the `cpu_baseline`
    object (the C oracle on the host cores, proof hash equal to the GPU's)."""
    import json
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--log-n", "12", "--steps", "4", "--warmup", "1", "--inflight", "4",
                        "--cpu-baseline-seconds", "2"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = lines[0]
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert d["metric"] == "proofs/sec" and d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 4 * 4 / (d["ms_per_step"] * 4 / 1e3)) / d["value"] < 0.01          # value = proofs of the timed steps / their time
    rf = d["roofline"]
    assert rf["bound"] in ("hbm", "mfma", "valu") and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and rf["kernel"] == "msm_accumulate_kernel"
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-5 and (rf["traffic"] is None or rf["traffic"] > 0)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import check_bench_line
    check_bench_line(d)          # every derived figure of the line recomputed from the line's own inputs (the CPU tier does the same on the committed lines)
    assert abs(rf["achieved"] - rf["pairs_per_launch"] * rf["algorithmic_bytes_per_pair"] / (rf["avg_launch_ms"] * 1e-3) / 1e9) / rf["achieved"] < 0.01
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["unit"] == "proofs/sec" and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
    assert cb["matches_gpu_proof"] is True
    assert len(d["round_ms"]) == 4 and d["witness_bits"]["proof_verifies"] is True


@pytest.mark.gpu
def test_rccl_branch_runs_on_hardware_world_1(gpu):
    """The RCCL data plane needs one GPU per rank, and these boxes have one GPU: a world-1 RCCL communicator is how
    ncclGetUniqueId / ncclCommInitRank / grouped ncclSend + ncclRecv on libapk's non-blocking stream / ncclAllGather /
    ncclCommCount / ncclCommDestroy execute on hardware at all (apk_comm_rccl_selftest; results compared byte for byte inside)."""
    sys.path.insert(0, ROOT)
    from algoplonk_amd import parallel
    assert parallel.rccl_selftest(0) == 1
    assert parallel.rccl_selftest(0) == 1          # init and teardown are repeatable in one process


@pytest.mark.gpu
@pytest.mark.parametrize("cname,G,log_n,bsb", [("bn254", 2, 9, 0), ("bn254", 4, 9, 0), ("bn254", 8, 9, 0), ("bls12-381", 8, 8, 0), ("bls12-381", 4, 10, 0),
                                               ("bls12-381", 8, 9, 1), ("bn254", 4, 8, 1)])
def test_subcoset_split_of_round_3_is_byte_identical(gpu, cname, G, log_n, bsb):
    """SURVEY.md section 8e row 2 / DESIGN.md section 6: rank k of G evaluates the wire polynomials on the points i = k (mod G) of
    the 4n coset only, runs the quotient kernel there, inverse-transforms locally; ONE all-gather and the last log2(G) stages
    give the quotient's coefficients.  G contexts in ONE process stand for the G ranks (G threads prove at the same time, the
    gather hook is a barrier + device-to-device copies between the contexts): every "rank" must return the proof of the
    whole-coset path, byte for byte - including G = 8, where Z(omega X) lives in another class than Z(X) and n + 3 coefficients
    fold onto n / 2 points."""
    import threading
    from algoplonk_amd import MarshalProof, _lib, plonk as ap_plonk, setup as ap_setup
    from algoplonk_amd._lib import lib, check
    from helpers import CURVES, blinding, random_chain_ccs
    from oracle.prng import tau_from_seed
    cv, ov = CURVES[cname]
    kw = {}
    if bsb:      # a BSB22 commitment: the committed column's evaluations (pi2) live on the sub-coset too
        from algoplonk_amd import workloads
        ccs, w, bl, tau = workloads.random_circuit_bsb22(cv, log_n, 0xA193)
        srs = ap_setup.unsafe_srs(cv, ccs.domain_size(), tau, device=gpu, lagrange=True)
        kw = {"hiding": [(0xA193, 0x3910A)]}
    else:
        ccs, w, sol = random_chain_ccs(cv, log_n, 0xA190 + log_n)
        srs = ap_setup.unsafe_srs(cv, ccs.domain_size(), tau_from_seed(5, cv.r), device=gpu)
        bl = blinding(cv, 9)
    pks = [ap_plonk.Setup(ccs, srs, device=gpu)[0] for _ in range(G)]
    plain = MarshalProof(ap_plonk.Prove(ccs, pks[0], w, bl, **kw))
    gate = threading.Barrier(G)
    bufs = {}
    HOOK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t)

    def make_hook(k):
        def hook(_user, d_all, nbytes):
            try:
                bufs[k] = d_all
                gate.wait(timeout=120)
                for r in range(G):
                    if r != k:
                        check(lib.apk_device_copy(pks[k].ctx, d_all + r * nbytes, bufs[r] + r * nbytes, nbytes))
                gate.wait(timeout=120)
                return 0
            except Exception:
                return _lib.APK_ERR_STATE
        return HOOK(hook)

    hooks = [make_hook(k) for k in range(G)]
    for k in range(G):
        check(lib.apk_ctx_set_subcoset(pks[k].ctx, k, G, hooks[k], None))
    out = [None] * G

    def run(k):
        try:
            out[k] = MarshalProof(ap_plonk.Prove(ccs, pks[k], w, bl, **kw))
        except Exception as e:
            out[k] = e
            gate.abort()

    ts = [threading.Thread(target=run, args=(k,)) for k in range(G)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert all(o == plain for o in out), [o if isinstance(o, Exception) else o == plain for o in out]
    for k in range(G):
        check(lib.apk_ctx_set_subcoset(pks[k].ctx, 0, 1, None, None))
    assert MarshalProof(ap_plonk.Prove(ccs, pks[1], w, bl, **kw)) == plain
    for pk in pks:
        pk.close()


@pytest.mark.gpu
def test_a_communicator_outlives_its_context(gpu):
    """a host with finalizers may destroy the context BEFORE the communicator that installed hooks on it (Python's
    teardown order, a Go finalizer).  apk_comm_destroy / apk_comm_bind then find the context gone in libapk's registry of live
    contexts and forget it instead of calling into freed memory; the communicator's device buffers go back to the runtime."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from algoplonk_amd import parallel, plonk as ap_plonk, setup as ap_setup, MarshalProof
    from helpers import CURVES, blinding, random_chain_ccs
    from oracle.prng import tau_from_seed
    cv, ov = CURVES["bn254"]
    ccs, w, sol = random_chain_ccs(cv, 8, 3)
    srs = ap_setup.unsafe_srs(cv, ccs.domain_size(), tau_from_seed(4, cv.r), device=gpu)
    for how in ("destroy", "rebind"):
        pk, vk = ap_plonk.Setup(ccs, srs, device=gpu)
        comm = parallel.Comm(0, 1)
        comm.bind(pk.ctx)
        comm.spmd_begin()
        assert len(MarshalProof(ap_plonk.Prove(ccs, pk, w, blinding(cv, 2)))) == 768
        pk.close()                                # the context goes first, hooks and all
        if how == "rebind":
            pk2, _ = ap_plonk.Setup(ccs, srs, device=gpu)
            comm.bind(pk2.ctx)                    # must not touch the dead context
            assert len(MarshalProof(ap_plonk.Prove(ccs, pk2, w, blinding(cv, 2)))) == 768
            comm.bind(None)
            pk2.close()
        comm.close()


@pytest.mark.gpu
def test_a_communicator_takes_its_hooks_off_the_context_when_it_goes(gpu):
    """apk_comm_spmd_begin installs hooks on the context whose user pointer is the communicator.  A communicator that is
    destroyed or rebound WITHOUT spmd_end (an exception between the two calls is enough) must take them off again - the next
    apk_prove on that context would otherwise call into freed memory.  World 1 here: the lifecycle, not the exchange."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from algoplonk_amd import parallel, plonk as ap_plonk, setup as ap_setup, MarshalProof
    from helpers import CURVES, blinding, random_chain_ccs
    from oracle.prng import tau_from_seed
    cv, ov = CURVES["bn254"]
    ccs, w, sol = random_chain_ccs(cv, 8, 3)
    srs = ap_setup.unsafe_srs(cv, ccs.domain_size(), tau_from_seed(4, cv.r), device=gpu)
    pk, vk = ap_plonk.Setup(ccs, srs, device=gpu)
    bl = blinding(cv, 2)
    want = MarshalProof(ap_plonk.Prove(ccs, pk, w, bl))
    for how in ("destroy", "rebind", "unbind"):
        comm = parallel.Comm(0, 1)
        comm.bind(pk.ctx)
        comm.spmd_begin()
        assert MarshalProof(ap_plonk.Prove(ccs, pk, w, bl)) == want          # through the hook (world 1: this rank commits everything)
        if how == "destroy":
            comm.close()                                                       # no spmd_end
        elif how == "rebind":
            comm.bind(pk.ctx)                                                  # binding again ends what the old binding installed
        else:
            comm.bind(None)
        assert MarshalProof(ap_plonk.Prove(ccs, pk, w, bl)) == want          # no hook left behind: the context proves on its own
        if how != "destroy":
            comm.close()
        assert MarshalProof(ap_plonk.Prove(ccs, pk, w, bl)) == want
    pk.close()
