#!/usr/bin/env python3
"""bench.py - BASELINE.json's metric on BASELINE.json's config.

metric   : proofs/sec (end-to-end apk_prove, witness already solved, SRS/trace/wire columns resident in HBM)
workload : configs[1] = BN254 random circuit, 2^17 constraints, synthetic SRS (seed 0xA190; the PPoT pk.bin is
           not in the mount - SURVEY.md §0.7), 1 MI355X per rank
step     : one batch of `--inflight` independent proofs of that circuit, proved concurrently on the context's
           slots (one host thread + one HIP stream per proof)
N > 1    : one process per GPU, every rank proves its own independent proofs (no data-path collective:
           "replicas only" for proofs) -> weak scaling; the only torch.distributed traffic is the
           timing barrier / MAX reduction the contract asks for.

Extra objects on the JSON line (prompt ④): `roofline` for the dominant kernel (msm_accumulate_kernel, HIP events on
the stream it runs on, algorithmic bytes = 96 B per (scalar, point) pair) and `cpu_baseline` (the C oracle timed on
this box's host cores on the same workload).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# one hardware queue per in-flight proof stream (the ROCm default of 4 serialises 16 streams onto 4 queues:
# 172 -> 205 proofs/s measured with 16; 24 queues / 24 proofs in flight is the plateau); set before HIP initialises
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log-n", type=int, default=17)
    ap.add_argument("--curve", default="bn254", choices=["bn254", "bls12_381"])
    ap.add_argument("--inflight", type=int, default=32,
                    help="independent proofs per step = concurrent apk_prove callers (the context runs 16 at a time, the rest wait for a slot)")
    ap.add_argument("--msm-window", type=int, default=0)
    ap.add_argument("--cpu-baseline-seconds", type=float, default=20.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mode", default="prove", choices=["prove", "msm-sharded"],
                    help="prove (default, BASELINE configs[1]) | msm-sharded (configs[3]: ONE MSM split by index range over the ranks)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    import torch  # device plumbing + torch.distributed only
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from algoplonk_amd import _lib, ecc, frontend, plonk, setup, workloads
    from algoplonk_amd._lib import lib, check

    cv = ecc.BN254 if args.curve == "bn254" else ecc.BLS12_381
    seed = 0xA190 if cv is ecc.BN254 else 0xA191
    if args.mode == "msm-sharded":
        return bench_sharded_msm(args, cv, rank, local_rank, world, torch, dist)
    t0 = time.time()
    wl = workloads.random_circuit(cv, args.log_n, seed)
    n = wl.ccs.domain_size()
    srs = setup.unsafe_srs(cv, n, wl.tau, device=local_rank)
    pk, vk = plonk.Setup(wl.ccs, srs, device=local_rank, msm_window=args.msm_window, slots=args.inflight)
    L, R, O = frontend.wire_columns(wl.ccs, wl.solution)
    host = [cv.fr_vector(v) for v in (L, R, O)]
    dptr = []
    for b in host:
        p = C.c_void_p()
        check(lib.apk_device_alloc(pk.ctx, len(b), C.byref(p)))
        check(lib.apk_device_upload(pk.ctx, p, b, len(b)))
        dptr.append(p)
    pub = cv.fr_vector(wl.witness.public)
    bl = cv.fr_vector(wl.blinding)
    setup_s = time.time() - t0

    proofs = [_lib.Proof() for _ in range(args.inflight)]
    errors = []

    def one(i):
        rc = lib.apk_prove_device(pk.ctx, dptr[0], dptr[1], dptr[2], pub, bl, None, C.byref(proofs[i]))
        if rc != 0:
            errors.append((rc, lib.apk_last_error()))

    def step():
        if args.inflight == 1:
            one(0)
            return
        ts = [threading.Thread(target=one, args=(i,)) for i in range(args.inflight)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t1
    if errors:
        raise SystemExit("apk_prove failed: %r" % (errors[0],))
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    total_proofs = args.steps * args.inflight * world
    value = total_proofs / elapsed

    # ---- single-proof latency + live HIP-event timing of the dominant kernel (own pass, after the timed region)
    pk.enable_stats(True)
    pk.stats(reset=True)
    lat0 = time.perf_counter()
    nlat = 3
    for _ in range(nlat):
        one(0)
    lat_ms = (time.perf_counter() - lat0) / nlat * 1e3
    st = pk.stats(reset=True)
    pk.enable_stats(False)
    pair_bytes = 32 + 2 * cv.fp_bytes  # 96 B/pair BN254, 128 B/pair BLS12-381
    acc_avg_ms = st.msm_accumulate_ms / max(st.msm_accumulate_launches, 1)
    pairs_per_launch = st.msm_pairs / max(st.msm_accumulate_launches, 1)
    achieved = pairs_per_launch * pair_bytes / (acc_avg_ms * 1e-3) / 1e9 if acc_avg_ms > 0 else 0.0
    # HBM bytes per launch from the committed PMC run (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes,
    # corrected as MI355X_MICROARCH.md prescribes; profiles/r01_pmc_msm_accumulate.json): bytes per pair x pairs
    traffic = None
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_msm_accumulate.json")))
        if cv is ecc.BN254 and args.log_n == 17:
            traffic = int(pmc["hbm_bytes_per_pair"] * pairs_per_launch)
    except Exception:
        traffic = None
    roofline = {
        "bound": "hbm", "kernel": "msm_accumulate_kernel", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS,
        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
        "avg_launch_ms": round(acc_avg_ms, 4), "pairs_per_launch": round(pairs_per_launch, 1),
        "algorithmic_bytes_per_pair": pair_bytes,
    }
    # The HBM fraction above is what the contract asks for; the kernel's real ceiling is VALU issue.
    # BN254 only: 2335 VALU instructions (1467 v_mad_u64_u32) per mixed addition in the build's assembly = 9.6 k issue cycles
    # per wave at the measured per-instruction costs -> SIMDs * clock / 9.6 k * 64 lanes additions/s if no SIMD ever stalled.
    if cv is ecc.BN254 and acc_avg_ms > 0:
        c = args.msm_window or (16 if args.log_n >= 21 else min(15, max(8, args.log_n - 2)))
        windows = (254 + 1 + c - 1) // c
        adds_per_s = pairs_per_launch * windows / (acc_avg_ms * 1e-3)
        issue_bound = 1024 * 2.4e9 / 9600.0 * 64
        roofline["valu"] = {"mixed_additions_per_s": round(adds_per_s / 1e9, 3), "issue_bound": round(issue_bound / 1e9, 3),
                            "unit": "G additions/s", "frac": round(adds_per_s / issue_bound, 4)}

    # ---- MSM-only throughput (second half of BASELINE.json's metric): one 2^log_n MSM, scalars resident in HBM
    out_pt = C.create_string_buffer(2 * cv.fp_bytes)
    for _ in range(3):
        check(lib.apk_msm_g1_device(pk.ctx, 0, dptr[0], n, out_pt))
    torch.cuda.synchronize()
    reps = 20
    m0 = time.perf_counter()
    for _ in range(reps):
        check(lib.apk_msm_g1_device(pk.ctx, 0, dptr[0], n, out_pt))
    torch.cuda.synchronize()
    msm_s = (time.perf_counter() - m0) / reps
    msm_mscalar = n / msm_s / 1e6

    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            from bench_cpu import cpu_baseline_prove
            cpu_baseline = cpu_baseline_prove(wl, srs, args.cpu_baseline_seconds)
        except Exception as e:  # the baseline is reported, never required for the GPU number
            cpu_baseline = {"value": None, "unit": "proofs/sec", "cores": 0, "kind": "port", "sample": "unavailable: %s" % e}

    gpu_proof_sha = None
    if rank == 0:
        import hashlib
        from algoplonk_amd import MarshalProof
        gpu_proof_sha = hashlib.sha256(MarshalProof(plonk.Proof(cv, proofs[0]))).hexdigest()[:16]
        if cpu_baseline and cpu_baseline.get("proof_sha256_prefix"):
            cpu_baseline["matches_gpu_proof"] = cpu_baseline["proof_sha256_prefix"] == gpu_proof_sha
    if rank == 0:
        line = {
            "metric": "proofs/sec", "value": round(value, 4), "unit": "proofs/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u32x8 (Montgomery Fr/Fp)" if cv is ecc.BN254 else "u32x8 Fr / u32x12 Fp (Montgomery)",
            "data": "synthetic",
            "config": {"workload": wl.name, "log_n": args.log_n, "curve": cv.name, "proofs_per_step": args.inflight,
                       "srs": "synthetic tau=SHA256(seed)", "parallelism": "replicas x%d" % world},
            "proof_latency_ms": round(lat_ms, 3), "msm_mscalar_per_s": round(msm_mscalar, 3),
            "msm_ms": round(msm_s * 1e3, 4), "setup_s": round(setup_s, 2),
            "msm_batch_avg_ms": round(st.msm_total_ms / max(st.msm_batches, 1), 4),
            "ntt_ms_per_proof": round(st.ntt_ms / max(st.proofs, 1), 4),
            "proof_sha256_prefix": gpu_proof_sha, "roofline": roofline, "cpu_baseline": cpu_baseline,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


def bench_sharded_msm(args, cv, rank, local_rank, world, torch, dist):
    """BASELINE.json configs[3]: one 2^log_n MSM (seed 0xA192, uniform scalars, SRS-shaped points) sharded by index range:
    every rank keeps the windowed tables of its slice resident, computes a full partial sum, then ONE all-gather of a
    64/96-byte point per rank + world-1 host point additions (algoplonk_amd/parallel.py).  Strong scaling."""
    from algoplonk_amd import parallel, setup, workloads
    from algoplonk_amd._lib import lib, check

    n = 1 << args.log_n
    g = workloads.SplitMix64(0xA192)
    tau = workloads.tau_from_seed(0xA192, cv.r)
    srs = setup.unsafe_srs(cv, n, tau, device=local_rank)
    scalars = cv.fr_vector([g.fr(cv.r) for _ in range(n)])
    sm = parallel.ShardedMsm(cv, srs.g1[: n * 2 * cv.fp_bytes], device=local_rank, rank=rank, world=world, msm_window=args.msm_window)
    mine = scalars[sm.lo * 32: sm.hi * 32]
    d = C.c_void_p()
    check(lib.apk_device_alloc(sm._ctx, len(mine), C.byref(d)))
    check(lib.apk_device_upload(sm._ctx, d, mine, len(mine)))
    out = C.create_string_buffer(2 * cv.fp_bytes)

    def step():
        check(lib.apk_msm_g1_device(sm._ctx, 0, d, sm.hi - sm.lo, out))
        return parallel.gather_and_add(cv, out.raw) if world > 1 else out.raw

    for _ in range(args.warmup):
        res = step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t1
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        import hashlib
        print(json.dumps({
            "metric": "MSM Mscalar/s", "value": round(n * args.steps / elapsed / 1e6, 3), "unit": "Mscalar/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "u29x9 Fp (BN254) / u28x14 Fp (BLS12-381) unsaturated Montgomery",
            "data": "synthetic", "config": {"workload": "%s single MSM 2^%d sharded by index range" % (cv.name, args.log_n),
                                            "parallelism": "index-range x%d + all-gather of %d-byte points" % (world, 2 * cv.fp_bytes)},
            "result_sha256_prefix": hashlib.sha256(res).hexdigest()[:16]}), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
