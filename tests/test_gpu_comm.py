"""GPU tier: libapk's communicator with real contexts.  The GPU boxes of this tier have ONE MI355X, so two (and three) ranks
share device 0: RCCL refuses two ranks on one device, the communicator notices (apk_comm_bind compares the ranks' device
ordinals) and stages the data plane through the host and its TCP star - every other line of csrc/comm.cpp (the dealing,
apk_msm_g1_batch_device over dealt index ranges, the per-wire dealing through apk_coset_ntt_device, the worker loop, the commit
and wire hooks inside apk_prove) is what an 8-GPU node runs."""
import ctypes as C
import multiprocessing as mp
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q, cname, split_wires):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    try:
        if split_wires:
            os.environ["APK_SPLIT_WIRES"] = "1"
        from algoplonk_amd import MarshalProof, _lib, parallel, plonk as ap_plonk, setup as ap_setup
        from helpers import CURVES, blinding, random_chain_ccs
        from oracle import plonk as oplonk
        from oracle.prng import SplitMix64, tau_from_seed
        cv, ov = CURVES[cname]
        comm = parallel.Comm(rank, world, "127.0.0.1", port)
        # ---- ONE proof, its commitments (and wires) dealt to the ranks: every rank holds the circuit context
        ccs, w, sol = random_chain_ccs(cv, 10, 0xA190 + 10)
        tau = tau_from_seed(99, cv.r)
        srs = ap_setup.unsafe_srs(cv, ccs.domain_size(), tau, device=0)
        pk, vk = ap_plonk.Setup(ccs, srs, device=0)
        comm.bind(pk.ctx)
        assert comm.transport == "tcp"                # two ranks on one device: host-staged data plane
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
        q.put((rank, "ok"))
    except Exception as e:
        import traceback
        q.put((rank, "FAIL: %r %s" % (e, traceback.format_exc()[-900:])))


@pytest.mark.gpu
@pytest.mark.parametrize("cname,world,split_wires", [("bn254", 2, False), ("bls12-381", 2, True), ("bn254", 3, True)])
def test_split_proof_and_sharded_msm_between_processes_sharing_the_gpu(gpu, cname, world, split_wires):
    from algoplonk_amd.parallel import free_port
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, cname, split_wires)) for r in range(world)]
    [p.start() for p in procs]
    res = [q.get(timeout=600) for _ in procs]
    [p.join(timeout=120) for p in procs]
    assert sorted(res) == [(r, "ok") for r in range(world)], res


@pytest.mark.gpu
@pytest.mark.parametrize("mode,extra", [("prove-split", ["--curve", "bls12_381", "--log-n", "12"]), ("msm-sharded", ["--log-n", "12"]),
                                        ("prove", ["--log-n", "12", "--inflight", "2"])])
def test_bench_py_runs_its_multi_rank_modes_under_the_contract_launcher(gpu, mode, extra):
    """`python bench.py --gpus 2 ...` re-executes itself under torch.distributed.run (the contract's command line), the two ranks
    rendezvous on libapk's communicator and rank 0 prints ONE JSON line with n_gpus = 2.  Functional only: both ranks share the
    box's one GPU (APK_BENCH_SHARE_GPU=1), so the numbers mean nothing."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["APK_BENCH_SHARE_GPU"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--mode", mode, "--steps", "3", "--warmup", "1",
                        "--no-pmc", "--no-cpu-baseline"] + extra, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = lines[0]
    assert line["n_gpus"] == 2 and line["config"]["world_size"] == 2 and line["steps"] == 3
    if mode == "prove-split":
        assert line["matches_single_gpu_proof"] is True and line["scaling"] == "strong"
    if mode == "prove":
        assert line["scaling"] == "weak" and line["value"] > 0
