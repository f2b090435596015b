#!/usr/bin/env python3
"""bench.py - BASELINE.json's metric on BASELINE.json's config.

metric   : proofs/sec (end-to-end apk_prove, witness already solved, SRS/trace/wire columns resident in HBM)
workload : configs[1] = BN254 random circuit, 2^17 constraints, synthetic SRS (seed 0xA190; the PPoT pk.bin is
           not in the mount - SURVEY.md §0.7), 1 MI355X per rank
step     : one batch of `--inflight` independent proofs of that circuit, proved concurrently on the context's
           slots (one host thread + one HIP stream per proof)

Launch contract:
  * `python bench.py --gpus N ...` with no WORLD_SIZE in the environment and N > 1 re-executes ITSELF as
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...`
    (one rank per GPU over RCCL) and relays rank 0's JSON line;
  * launched by that command (or by the driver's own torchrun) it reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the
    environment; `--gpus` must then equal WORLD_SIZE.
Modes:
  prove (default)  : every rank proves its own independent proofs - "replicas only", no data-path collective, weak scaling
                     (BASELINE configs[1], and configs[4] with --curve bls12_381 --log-n 21 --bsb22 1);
  msm-sharded      : configs[3] - ONE 2^log_n MSM split by index range over the ranks, one all-gather of a 64/96-byte point
                     per rank + local point additions; strong scaling;
  prove-split      : intra-proof multi-GPU: ONE proof at a time, its commitment batches dealt to the
                     ranks by index range (algoplonk_amd/parallel.py::SplitCommitter: scatter of scalar slices, per-rank
                     partial MSMs, all-gather of the partial sums), transcript on rank 0; strong scaling - meant for
                     --curve bls12_381 --log-n 21;
  prove-spmd       : the same proof on every rank (replicated prover, round 4): every rank holds context + witness and proves;
                     only the commitments are shared out by index range from each rank's own polynomials (one all-gather of
                     partial sums per batch, nothing scattered); strong scaling;
  launcher-selftest: NOT a measurement - the launcher, rendezvous, barrier / MAX reduction and JSON plumbing with a no-op
                     step (CPU tier test of this file).

Extra objects on the JSON line (prompt ④): `roofline` for the dominant kernel (msm_accumulate_kernel, HIP events on the
stream it runs on, algorithmic bytes = 96 B per (scalar, point) pair on BN254; `traffic` = HBM bytes per launch from
rocprofv3 PMC passes run by this script) and `cpu_baseline` (the C oracle timed on this box's host cores on the same
workload, plus the result of probing for a Go toolchain that could run the real gnark prover: bench/gnark_cpu).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import shutil
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# one hardware queue per in-flight proof stream (libapk sets the same default when it is loaded; set here too because torch
# may initialise HIP first: the ROCm default of 4 serialises 16 streams onto 4 queues, 172 -> 205 proofs/s measured with 16)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--log-n", type=int, default=17)
    ap.add_argument("--curve", default="bn254", choices=["bn254", "bls12_381"])
    ap.add_argument("--bsb22", type=int, default=0, help="BSB22 commitments in the circuit (configs[4]: 1)")
    ap.add_argument("--inflight", type=int, default=32,
                    help="independent proofs per step = concurrent apk_prove callers (the context runs 16 at a time, the rest wait for a slot)")
    ap.add_argument("--msm-window", type=int, default=0)
    ap.add_argument("--cpu-baseline-seconds", type=float, default=20.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 PMC passes behind roofline.traffic")
    ap.add_argument("--witness", default="uniform", choices=["uniform", "bits"],
                    help="wire values of the synthetic circuit: uniform Fr (BASELINE.md section 2) or the bit-heavy circuit of "
                         "workloads.skewed_circuit; the default run reports the bit-heavy rate beside the headline (`witness_bits`)")
    ap.add_argument("--witnesses", type=int, default=0,
                    help="distinct assignments of the circuit the concurrent callers prove (every caller walks through all of them; "
                         "default 16 up to 2^18, 8 at 2^19, 2 from 2^20 - generating an assignment of a 2^21 circuit takes the Python harness half a minute; never more than --inflight)")
    ap.add_argument("--no-oracle-check", action="store_true",
                    help="skip the per-witness comparison of the timed region's proofs with the C oracle's (bench_cpu.oracle_blobs)")
    ap.add_argument("--no-host-inputs", action="store_true", help="skip the apk_prove (host pointers) legs")
    ap.add_argument("--step-barrier", action="store_true",
                    help="join all callers after every step (rounds 1-2); default: persistent callers, barriers only around the K steps")
    ap.add_argument("--mode", default="prove", choices=["prove", "msm-sharded", "prove-split", "prove-spmd", "launcher-selftest"])
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------- launcher
def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_command(argv, gpus: int, port: int):
    """The command line the contract prescribes for N > 1 (one rank per GPU on ONE node)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.join(ROOT, "bench.py")] + list(argv)


def maybe_self_launch(args, argv) -> bool:
    """`--gpus N` outside torchrun: re-execute under torch.distributed.run.  Returns True when this process was the launcher."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return False
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this pool (RCCL needs it)
    cmd = launch_command(argv, args.gpus, free_port())
    r = subprocess.run(cmd, env=env)
    if r.returncode != 0:
        raise SystemExit("bench.py launcher: %s exited with %d" % (" ".join(cmd[:8]), r.returncode))
    return True


class Ranks:
    """RANK / LOCAL_RANK / WORLD_SIZE + the barrier / MAX-over-ranks timing of the contract, on libapk's OWN communicator
    (include/apk.h apk_comm_*: TCP control plane, RCCL data plane) - no torch in the process, hence no second HIP runtime.
    The launcher (torchrun) only exports the environment.  Every library call of a step returns its result to the host, so the
    device is idle when a step returns: the barrier alone brackets the timed region the way barrier + synchronize would."""

    def __init__(self, args, need_gpu: bool = True):
        from algoplonk_amd import _lib, parallel
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if os.environ.get("APK_BENCH_SHARE_GPU") == "1":
            self.local_rank = 0       # functional runs of the N > 1 modes on a one-GPU box: every rank on device 0 (NOT a measurement)
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if self.world != args.gpus:
            raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, self.world))
        if need_gpu:
            ndev = _lib.device_count()
            if ndev == 0:
                raise SystemExit("bench.py needs an MI355X: no HIP device visible (the HIP path has no CPU fallback)")
            if self.local_rank >= ndev:
                raise SystemExit("bench.py: LOCAL_RANK %d but only %d HIP device(s) visible" % (self.local_rank, ndev))
        self.comm = parallel.Comm.from_env()
        self.backend = "libapk comm"

    def fence(self):
        self.comm.barrier()

    def timed(self, step, steps: int, warmup: int) -> float:
        """W untimed steps, then EXACTLY K steps bracketed by barriers; MAX over ranks."""
        for _ in range(warmup):
            step()
        self.fence()
        t1 = time.perf_counter()
        for _ in range(steps):
            step()
        self.fence()
        return self.comm.max(time.perf_counter() - t1)

    def timed_callers(self, one, callers: int, steps: int, warmup: int, step_barrier: bool = False) -> float:
        """The same contract with `callers` PERSISTENT host threads (a proving service's request handlers): a step is one call of
        `one(i)` by every caller i, the timed region holds exactly steps x callers calls between the two barriers.  Nothing waits
        BETWEEN steps - a caller that is done with its proof of step k starts its proof of step k + 1 - so the proving slots do
        not drain and refill 60 times inside the timed region (`step_barrier` joins every step like round 1-2 did)."""
        if callers == 1 or step_barrier:
            def step():
                if callers == 1:
                    one(0)
                    return
                ts = [threading.Thread(target=one, args=(i,)) for i in range(callers)]
                for t in ts:
                    t.start()
                for t in ts:
                    t.join()
            return self.timed(step, steps, warmup)
        gate = threading.Barrier(callers + 1)

        def run(i):
            for _ in range(warmup):
                one(i)
            gate.wait()          # every caller's warm-up is done
            gate.wait()          # released together once the ranks have met
            for _ in range(steps):
                one(i)

        ts = [threading.Thread(target=run, args=(i,)) for i in range(callers)]
        for t in ts:
            t.start()
        gate.wait()
        self.fence()
        t1 = time.perf_counter()
        gate.wait()
        for t in ts:
            t.join()
        self.fence()
        return self.comm.max(time.perf_counter() - t1)

    def data_plane_probe(self, ctx, timeout_s: float = 120.0) -> dict:
        """Independent proofs exchange no data, so the timed region of `prove` mode never touches the data plane.  AFTER the
        measurements every rank binds the communicator to its context once (collective): RCCL when every rank owns a GPU
        (ncclCommInitRank + one ncclAllGather of the rank numbers, checked), else HIP IPC, else TCP - and the line reports what
        came up (`rccl_ranks` = ncclCommCount).  A watchdog bounds it: a bind that does not return is reported as such and the
        process leaves without the library's teardown (the line is already complete)."""
        if self.world == 1:
            return {"transport": "none (single process)", "rccl_ranks": 0}
        res = {}

        def go():
            try:
                self.comm.bind(ctx)
                res.update(self.plane_report())
                self.comm.bind(None)
            except Exception as e:  # reported, never fatal for a line that is already measured
                res["transport"], res["rccl_ranks"], res["error"] = "unavailable", 0, str(e)[:200]

        t = threading.Thread(target=go, daemon=True)
        t.start()
        t.join(timeout_s)
        if t.is_alive():
            self.hung = True
            return {"transport": "unknown", "rccl_ranks": 0, "error": "apk_comm_bind did not return within %d s" % timeout_s}
        return res

    hung = False

    def plane_report(self) -> dict:
        """What the data plane came up as, measured (collective: every rank calls it after apk_comm_bind).  RCCL on every rank ->
        "rccl"; anything else is reported as "fallback:<transport>:<why>" so that a line timed over the TCP star or IPC cannot be
        mistaken for an xGMI measurement.  link_gbps: one timed 64 MiB ncclSend/ncclRecv ring + one 32 MiB-per-rank all-gather
        (apk_comm_link_probe) - the per-link rate DESIGN.md's multi-GPU projections assume (48 GB/s) and no box of the build could
        measure."""
        t, n = self.comm.transport, self.comm.rccl_ranks
        rep = {"transport": t if (t == "rccl" and n == self.world) else "fallback:%s:%s" % (t, self.comm.transport_reason or "?"),
               "rccl_ranks": n}
        if self.world > 1:
            try:
                rep["link_gbps"] = self.comm.link_probe()
            except Exception as e:
                rep["link_gbps"] = {"error": str(e)[:160]}
        return rep

    def close(self):
        if self.hung:
            sys.stdout.flush()
            os._exit(0)
        self.comm.close()


# ---------------------------------------------------------------------------------------------------- probes
def go_probe() -> dict:
    """is there a Go toolchain (and a module cache with gnark) that could run bench/gnark_cpu - the REAL
    reference CPU prover?  Recorded in cpu_baseline; when it succeeds the gnark number replaces the port's."""
    go = shutil.which("go")
    if not go:
        return {"go": None, "gnark_cpu": "not run: no Go toolchain on this box (go: command not found)"}
    try:
        ver = subprocess.run([go, "version"], capture_output=True, text=True, timeout=20).stdout.strip()
    except Exception as e:
        return {"go": "error: %s" % e, "gnark_cpu": "not run"}
    return {"go": ver, "gnark_cpu": "untried"}


def gnark_cpu_baseline(probe: dict, curve: str, log_n: int, budget_s: float):
    """build and run bench/gnark_cpu (gnark v0.15.0 plonk.Prove on the same circuit shape).  Needs the Go
    modules in the module cache (no network here); any failure is reported, never fatal."""
    if not probe.get("go") or probe["gnark_cpu"] != "untried":
        return None
    src = os.path.join(ROOT, "bench", "gnark_cpu")
    env = dict(os.environ, GOFLAGS="-mod=mod", GOPROXY="off")
    try:
        b = subprocess.run(["go", "build", "-o", os.path.join(ROOT, "oracle", "_ref", "gnark_cpu"), "."], cwd=src, env=env,
                           capture_output=True, text=True, timeout=300)
        if b.returncode != 0:
            probe["gnark_cpu"] = "go build failed: %s" % (b.stderr.strip().splitlines() or ["?"])[-1][:200]
            return None
        r = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "gnark_cpu"), "-curve", curve, "-log-n", str(log_n), "-seconds", str(budget_s)],
                           capture_output=True, text=True, timeout=budget_s * 4 + 600)
        line = json.loads(r.stdout.strip().splitlines()[-1])
        probe["gnark_cpu"] = "ran"
        return {"value": line["proofs_per_sec"], "unit": "proofs/sec", "cores": line["cores"], "kind": "reference",
                "sample": "%d gnark v0.15.0 plonk.Prove runs of a 2^%d %s circuit in %.1f s (bench/gnark_cpu)" % (line["proofs"], log_n, curve, line["seconds"])}
    except Exception as e:
        probe["gnark_cpu"] = "failed: %s" % str(e)[:200]
        return None


def valu_issue_rate():
    """Wall time per wave instruction and SIMD of a dependent v_mad_u64_u32 chain, measured NOW on this box by
    tools/ubench/valu_rates --json (built by __graft_entry__.build()): the issue cost the accumulate kernel's instruction count is
    priced at.  None when the binary is missing or fails - the line then carries no issue bound rather than a stale one."""
    exe = os.path.join(ROOT, "tools", "ubench", "valu_rates")
    if not os.path.exists(exe):
        return None
    try:
        r = subprocess.run([exe, "--json"], capture_output=True, text=True, timeout=60)
        return json.loads(r.stdout.strip().splitlines()[-1])
    except Exception:
        return None


def pmc_traffic(curve: str, log_n: int, window: int, timeout_s: float = 420.0):
    """HBM bytes per (scalar, point) pair and VALU instructions per bucket addition of msm_accumulate_kernel, measured NOW on
    this box: three separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ_INSTS_VALU - one counter per pass and no trace
    domains, as MI355X_MICROARCH.md's HBM section prescribes) over tools/prof_msm.py (4 single MSMs + the 8 trace commitments of
    Setup).  gfx950 correction from the same guide: FETCH_SIZE counts 64-byte units reported in KB at half weight -> doubled;
    WRITE_SIZE as is.  None when rocprofv3 is missing or a traffic pass fails; the instruction pass is optional."""
    if not shutil.which("rocprofv3"):
        return None
    import sqlite3
    import tempfile
    out = {}
    env = dict(os.environ, TMPDIR="/tmp")
    if window:
        env["APK_MSM_WINDOW"] = str(window)
    # (the HBM counters one per pass, as the guide prescribes; the three SQ instruction counters share a pass)
    for counter in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32"):
        d = tempfile.mkdtemp(prefix="apk_pmc_", dir="/tmp")
        try:
            subprocess.run(["rocprofv3", "--kernel-trace", "--pmc"] + counter.split() + ["-d", d, "-o", "p", "--", sys.executable,
                            os.path.join(ROOT, "tools", "prof_msm.py"), str(log_n), "4", "0", curve], cwd="/tmp", env=env, capture_output=True,
                           timeout=timeout_s, check=True)
            dbs = [os.path.join(r, f) for r, _, fs in os.walk(d) for f in fs if f.endswith("_results.db")]
            db = sqlite3.connect(dbs[0])
            tots, launches = {}, set()
            for name, did, cname, val in db.execute("select kernel_name, dispatch_id, counter_name, value from counters_collection"):
                if "msm_accumulate_kernel" in name and cname in counter.split():
                    tots[cname] = tots.get(cname, 0.0) + val
                    launches.add(did)
            for cname in counter.split():
                if cname in tots:
                    out[cname] = (tots[cname], len(launches))
        except Exception:
            if counter.startswith("SQ_"):
                continue
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    if "FETCH_SIZE" not in out or "WRITE_SIZE" not in out:
        return None
    n_msm = 12                                     # prof_msm.py: 4 single MSMs + 8 VK commitments (two batches of 4)
    pairs = n_msm * (1 << log_n)
    fetch_kb, nl = out["FETCH_SIZE"]
    write_kb, _ = out["WRITE_SIZE"]
    bytes_total = 2.0 * fetch_kb * 1024.0 + write_kb * 1024.0
    res = {"hbm_bytes_per_pair": bytes_total / pairs, "launches": nl, "fetch_kb": fetch_kb, "write_kb": write_kb, "window_bits": window}
    if "SQ_INSTS_VALU" in out and window:
        # SQ_INSTS_VALU counts wave instructions; a bucket addition is one lane's work: x 64 lanes / (pairs x windows) additions
        from algoplonk_amd import ecc
        rbits = (ecc.BN254 if curve == "bn254" else ecc.BLS12_381).r.bit_length()
        windows = (rbits + 1 + window - 1) // window
        res["valu_wave_instructions"] = out["SQ_INSTS_VALU"][0]
        res["valu_instructions_per_addition"] = out["SQ_INSTS_VALU"][0] * 64.0 / (pairs * windows)
        if "SQ_INSTS_VALU_INT64" in out:          # the dynamic instruction mix: v_mad_u64_u32 and the 64-bit shifts / adds
            res["valu_int64_share"] = out["SQ_INSTS_VALU_INT64"][0] / max(out["SQ_INSTS_VALU"][0], 1.0)
            res["valu_int32_share"] = out.get("SQ_INSTS_VALU_INT32", (0.0, 0))[0] / max(out["SQ_INSTS_VALU"][0], 1.0)
    return res


def pmc_valu_per_proof(curve: str, log_n: int, callers: int, bsb22: int, timeout_s: float = 300.0):
    """VALU wave instructions of ONE proof made under load (the loaded forms of the kernels, gangs included), measured NOW: two
    rocprofv3 --pmc SQ_INSTS_VALU runs of tools/prof_loaded_proofs.py that differ by 2 x callers proofs; the difference of their
    totals over every kernel / the proofs in between.  None when rocprofv3 is missing or a run fails."""
    if not shutil.which("rocprofv3"):
        return None
    import sqlite3
    import tempfile
    env = dict(os.environ, TMPDIR="/tmp")
    tot = {}
    for rounds in (1, 3):
        d = tempfile.mkdtemp(prefix="apk_pmcv_", dir="/tmp")
        try:
            subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", "SQ_INSTS_VALU", "-d", d, "-o", "p", "--", sys.executable,
                            os.path.join(ROOT, "tools", "prof_loaded_proofs.py"), curve, str(log_n), str(callers), str(rounds), str(bsb22)],
                           cwd="/tmp", env=env, capture_output=True, timeout=timeout_s, check=True)
            dbs = [os.path.join(r, f) for r, _, fs in os.walk(d) for f in fs if f.endswith("_results.db")]
            db = sqlite3.connect(dbs[0])
            tot[rounds] = sum(v for (v,) in db.execute("select value from counters_collection where counter_name = 'SQ_INSTS_VALU'"))
        except Exception:
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    proofs = 2 * callers
    return {"wave_instructions_per_proof": (tot[3] - tot[1]) / proofs, "proofs_between_the_runs": proofs, "callers": callers}


# ---------------------------------------------------------------------------------------------------- modes
def main(argv=None) -> None:
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if maybe_self_launch(args, argv):
        return
    if args.mode == "launcher-selftest":
        return launcher_selftest(args)
    rk = Ranks(args)
    from algoplonk_amd import ecc
    cv = ecc.BN254 if args.curve == "bn254" else ecc.BLS12_381
    if args.mode == "msm-sharded":
        return bench_sharded_msm(args, cv, rk)
    if args.mode == "prove-split":
        return bench_prove_split(args, cv, rk)
    if args.mode == "prove-spmd":
        return bench_prove_spmd(args, cv, rk)
    return bench_prove(args, cv, rk)


def bench_prove_spmd(args, cv, rk) -> None:
    """One proof at a time on N GPUs, replicated prover: EVERY rank holds the circuit context and the witness and runs the same
    apk_prove_device; only the commitments are shared out (rank r commits its index range of every batch from its own copy of the
    polynomials, one all-gather of the partial sums - apk_comm_spmd_begin, csrc/comm.cpp).  Nothing is scattered and no rank
    idles.  A step = one proof (the same proof on every rank); value = proofs/s of the whole job (strong)."""
    import hashlib
    from algoplonk_amd import frontend, plonk, setup, workloads, MarshalProof
    from algoplonk_amd._lib import lib, check
    from algoplonk_amd import _lib

    seed = 0xA190 if args.curve == "bn254" else 0xA193
    wl = workloads.random_circuit(cv, args.log_n, seed)
    n = wl.ccs.domain_size()
    srs = setup.unsafe_srs(cv, n, wl.tau, device=rk.local_rank)
    pk, vk = plonk.Setup(wl.ccs, srs, device=rk.local_rank, msm_window=args.msm_window, slots=1)
    rk.comm.bind(pk.ctx)
    plane = rk.plane_report()
    L, R, O = frontend.wire_columns(wl.ccs, wl.solution)
    dptr = []
    for v in (L, R, O):
        b = cv.fr_vector(v)
        p = C.c_void_p()
        check(lib.apk_device_alloc(pk.ctx, len(b), C.byref(p)))
        check(lib.apk_device_upload(pk.ctx, p, b, len(b)))
        dptr.append(p)
    pub, bl = cv.fr_vector(wl.witness.public), cv.fr_vector(wl.blinding)
    proof = _lib.Proof()

    def step():
        check(lib.apk_prove_device(pk.ctx, dptr[0], dptr[1], dptr[2], pub, bl, None, C.byref(proof)))

    step()                                    # single-GPU reference proof (no hook) for the byte comparison
    want = MarshalProof(plonk.Proof(cv, proof))
    rk.comm.spmd_begin()
    for _ in range(args.warmup):
        step()
    rk.comm.phase_ms(reset=True)
    elapsed = rk.timed(step, args.steps, 0)
    ph = rk.comm.phase_ms(reset=True)
    got = MarshalProof(plonk.Proof(cv, proof))
    rk.comm.spmd_end()
    # this rank's time per proof by phase (host wall clock): its share of the commitment MSMs, the exchanges, everything else
    # (transforms, quotient, grand product, openings, transcript) - to be read against DESIGN.md section 6's projection
    per = 1.0 / max(args.steps, 1)
    phases = {"msm_ms": round(ph["msm_ms"] * per, 3), "sums_exchange_ms": round(ph["sums_exchange_ms"] * per, 3),
              "subcoset_gather_ms": round(ph["subcoset_gather_ms"] * per, 3),
              "non_msm_ms": round(elapsed * 1e3 * per - (ph["msm_ms"] + ph["sums_exchange_ms"] + ph["subcoset_gather_ms"]) * per, 3),
              "commit_rounds_per_proof": round(ph["commit_rounds"] * per, 2), "subcoset_split": bool(ph["gathers"])}
    same = 1.0 if got == want else 0.0
    same = -rk.comm.max(-same)                # every rank must hold the single-GPU proof
    if rk.rank == 0:
        print(json.dumps({
            "metric": "proofs/sec", "value": round(args.steps / elapsed, 4), "unit": "proofs/sec", "n_gpus": rk.world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "u32x8 Fr / u32x12 Fp (Montgomery)" if cv.name != "bn254" else "u32x8 (Montgomery Fr/Fp)", "data": "synthetic",
            "config": {"workload": wl.name + ", one proof at a time", "log_n": args.log_n, "curve": cv.name,
                       "parallelism": "replicated prover x%d: commitments by index range from every rank's own polynomials (all-gather of partial sums), nothing scattered" % rk.world,
                       "world_size": rk.world, "backend": "libapk comm: tcp control plane, %s data plane" % rk.comm.transport,
                       "rccl_ranks": rk.comm.rccl_ranks, "data_plane": plane},
            "phase_ms_per_proof_rank0": phases,
            "proof_sha256_prefix": hashlib.sha256(got).hexdigest()[:16], "matches_single_gpu_proof": same == 1.0}), flush=True)
    rk.comm.bind(None)
    pk.close()
    rk.close()


def bench_prove_split(args, cv, rk) -> None:
    """One proof at a time on N GPUs: every rank holds the circuit context; rank 0 proves with its commitment batches (and,
    with APK_SPLIT_WIRES=1, its per-wire coset evaluations) dealt to the ranks by libapk's communicator, the other ranks serve
    (apk_comm_split_begin / apk_comm_serve, csrc/comm.cpp).  A step = one proof; value = proofs/s of the whole job (strong)."""
    import hashlib
    from algoplonk_amd import _lib, frontend, plonk, setup, workloads, MarshalProof
    from algoplonk_amd._lib import lib, check

    seed = 0xA190 if args.curve == "bn254" else 0xA193
    wl = workloads.random_circuit(cv, args.log_n, seed)
    n = wl.ccs.domain_size()
    srs = setup.unsafe_srs(cv, n, wl.tau, device=rk.local_rank)
    pk, vk = plonk.Setup(wl.ccs, srs, device=rk.local_rank, msm_window=args.msm_window, slots=1)
    rk.comm.bind(pk.ctx)
    plane = rk.plane_report()
    line = None
    if rk.rank == 0:
        L, R, O = frontend.wire_columns(wl.ccs, wl.solution)
        dptr = []
        for v in (L, R, O):
            b = cv.fr_vector(v)
            p = C.c_void_p()
            check(lib.apk_device_alloc(pk.ctx, len(b), C.byref(p)))
            check(lib.apk_device_upload(pk.ctx, p, b, len(b)))
            dptr.append(p)
        pub, bl = cv.fr_vector(wl.witness.public), cv.fr_vector(wl.blinding)
        proof = _lib.Proof()

        def step():
            check(lib.apk_prove_device(pk.ctx, dptr[0], dptr[1], dptr[2], pub, bl, None, C.byref(proof)))

        step()                                    # single-GPU reference proof (no hook) for the byte comparison
        want = MarshalProof(plonk.Proof(cv, proof))
        rk.comm.split_begin()
        for _ in range(args.warmup):
            step()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        elapsed = time.perf_counter() - t1
        got = MarshalProof(plonk.Proof(cv, proof))
        rk.comm.split_end()
        line = {
            "metric": "proofs/sec", "value": round(args.steps / elapsed, 4), "unit": "proofs/sec", "n_gpus": rk.world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "u32x8 Fr / u32x12 Fp (Montgomery)" if cv.name != "bn254" else "u32x8 (Montgomery Fr/Fp)", "data": "synthetic",
            "config": {"workload": wl.name + ", one proof at a time", "log_n": args.log_n, "curve": cv.name,
                       "parallelism": "commitment batches dealt by index range x%d (scatter + all-gather of partial sums)%s, transcript on rank 0"
                                      % (rk.world, ", wires dealt by polynomial" if os.environ.get("APK_SPLIT_WIRES") == "1" else ""),
                       "world_size": rk.world, "backend": "libapk comm: tcp control plane, %s data plane" % rk.comm.transport,
                       "rccl_ranks": rk.comm.rccl_ranks, "data_plane": plane},
            "proof_sha256_prefix": hashlib.sha256(got).hexdigest()[:16], "matches_single_gpu_proof": got == want,
        }
    else:
        rk.comm.serve()
    # the contract's barrier: the leader's clock covers every rank's work (workers only serve its steps)
    rk.fence()
    if rk.rank == 0:
        print(json.dumps(line), flush=True)
    rk.close()


def launcher_selftest(args) -> None:
    rk = Ranks(args, need_gpu=False)
    state = {"n": 0}

    def step():
        state["n"] += 1

    elapsed = rk.timed(step, args.steps, args.warmup)
    assert state["n"] == args.steps + args.warmup
    if rk.rank == 0:
        print(json.dumps({"metric": "launcher-selftest", "value": 0.0, "unit": "none", "n_gpus": rk.world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": round(elapsed / max(args.steps, 1) * 1e3, 6), "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": "none", "data": "none: no-op step, NOT a measurement",
                          "config": {"workload": "launcher self-test", "backend": rk.backend, "world_size": rk.world}}), flush=True)
    rk.close()


def window_bits(args) -> int:
    """the window the timed context used: asked of the library (apk_ctx_msm_window) by the mode and parked in args; the formula
    below only stands in before a context exists (backend_impl.h choose_window)"""
    if args.msm_window:
        return args.msm_window
    if getattr(args, "window_used", 0):
        return args.window_used
    throughput = args.mode == "prove" and args.inflight > 2
    if args.log_n >= 20:
        return 19 if args.log_n <= 21 else 18 if args.log_n <= 23 else 16
    if args.log_n in (18, 19) and args.curve == "bn254":
        return 17
    return 16 if (args.log_n >= 17 and throughput) else min(15, max(8, args.log_n - 2))


def roofline_from_stats(args, cv, st, pmc, issue=None):
    pair_bytes = 32 + 2 * cv.fp_bytes  # 96 B/pair BN254, 128 B/pair BLS12-381
    acc_avg_ms = st.msm_accumulate_ms / max(st.msm_accumulate_launches, 1)
    pairs_per_launch = st.msm_pairs / max(st.msm_accumulate_launches, 1)
    achieved = pairs_per_launch * pair_bytes / (acc_avg_ms * 1e-3) / 1e9 if acc_avg_ms > 0 else 0.0
    roofline = {
        # the contract's HBM yardstick (algorithmic bytes / time / 8 TB/s) - but the kernel's real ceiling is integer-VALU
        # issue, reported beside it as `valu`
        "bound": "valu", "kernel": "msm_accumulate_kernel", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS,
        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
        "traffic": int(pmc["hbm_bytes_per_pair"] * pairs_per_launch) if pmc else None,
        "traffic_source": ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes run by this bench.py invocation (%d launches), %.0f B per pair"
                           % (pmc["launches"], pmc["hbm_bytes_per_pair"])) if pmc else None,
        "avg_launch_ms": round(acc_avg_ms, 4), "pairs_per_launch": round(pairs_per_launch, 1),
        "algorithmic_bytes_per_pair": pair_bytes,
    }
    if acc_avg_ms > 0:
        c = window_bits(args)
        windows = (cv.r.bit_length() + 1 + c - 1) // c
        adds_per_s = pairs_per_launch * windows / (acc_avg_ms * 1e-3)
        roofline["valu"] = {"bucket_additions_per_s": round(adds_per_s / 1e9, 3), "unit": "G additions/s", "windows": windows,
                            "scope": "ONE msm_accumulate_kernel launch of a lone proof (HIP events): a small circuit's lone launch cannot fill the device; "
                                     "the loaded device against its issue rate is roofline.valu_under_load"}
        if pmc and pmc.get("valu_instructions_per_addition") and issue:
            # the kernel's real ceiling: VALU issue.  Both factors are of THIS run: VALU instructions per bucket addition and lane
            # from the SQ_INSTS_VALU pass above, wall time per wave instruction and SIMD from tools/ubench/valu_rates --json
            # (dependent v_mad_u64_u32 chain, 4 waves per SIMD), on (4 x CUs) SIMDs x 64 lanes
            ipa = pmc["valu_instructions_per_addition"]
            ns_mad = min(issue["mad_u64_dependent_ns_per_wave_inst_per_simd_4waves"], issue["mad_u64_dependent_ns_per_wave_inst_per_simd_8waves"])
            ns, mix = ns_mad, "every instruction priced as a v_mad_u64_u32"
            share64 = pmc.get("valu_int64_share")
            if share64 is not None and 0.3 <= share64 <= 0.95 and issue.get("mix_75pct_mad_ns_per_wave_inst_per_simd"):
                # priced at the rate of an INTERLEAVED stream with this run's share of 64-bit integer instructions (SQ_INSTS_VALU_INT64:
                # the mads and 64-bit shifts): mads and simple adds overlap in the pipeline, so the stream issues faster than the sum of
                # its classes - interpolated between the ubench's 50 %, 75 % and 100 % mad streams (best of 2 / 4 / 8 waves per SIMD)
                pts = [(0.5, issue["mix_50pct_mad_ns_per_wave_inst_per_simd"]), (0.75, issue["mix_75pct_mad_ns_per_wave_inst_per_simd"]), (1.0, ns_mad)]
                lo, hi = (pts[0], pts[1]) if share64 <= 0.75 else (pts[1], pts[2])
                ns = lo[1] + (hi[1] - lo[1]) * (share64 - lo[0]) / (hi[0] - lo[0])
                mix = "%.1f %% 64-bit integer instructions, priced at the ubench's interleaved mad / add streams" % (100 * share64)
            elif share64 is not None and 0.3 <= share64 <= 0.95 and issue.get("add_co_chain_ns_per_wave_inst_per_simd_8waves"):
                ns = share64 * ns_mad + (1.0 - share64) * issue["add_co_chain_ns_per_wave_inst_per_simd_8waves"]
                mix = "%.1f %% 64-bit integer instructions at the mad rate, the rest at the 32-bit add-chain rate" % (100 * share64)
            simds = 4 * (issue.get("compute_units") or 256)
            bound = simds * 64 / (ipa * ns * 1e-9)
            roofline["valu"].update({"instructions_per_addition": round(ipa, 1), "ns_per_wave_instruction_per_simd": round(ns, 4), "simds": simds,
                                     "issue_bound": round(bound / 1e9, 3), "frac": round(adds_per_s / bound, 4),
                                     "issue_rates": issue,
                                     "basis": "this run: SQ_INSTS_VALU pass x 64 lanes / (pairs x windows) additions; tools/ubench/valu_rates --json "
                                              "(dependent chains, the faster of 4 and 8 waves per SIMD); " + mix})
        if pmc:
            roofline["hbm_traffic_frac"] = round(pmc["hbm_bytes_per_pair"] * pairs_per_launch / (acc_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    return roofline


def host_cpu_snapshot() -> dict:
    """Process CPU time and the cgroup's throttling counters (cpu.stat): the GPU boxes grant 16 CPUs' worth of time per 100 ms
    period to a process that sees 256 - a process that burns more is frozen, ALL of its threads, until the next period."""
    t = os.times()
    snap = {"user_s": t.user, "sys_s": t.system}
    try:
        for ln in open("/sys/fs/cgroup/cpu.stat"):
            k, v = ln.split()
            if k in ("nr_periods", "nr_throttled", "throttled_usec"):
                snap[k] = int(v)
    except Exception:
        pass
    return snap


def host_cpu_delta(a: dict, b: dict, wall_s: float) -> dict:
    d = {"cpu_s": round(b["user_s"] - a["user_s"] + b["sys_s"] - a["sys_s"], 3), "wall_s": round(wall_s, 3)}
    d["cores_used"] = round(d["cpu_s"] / wall_s, 2) if wall_s > 0 else None
    if "nr_throttled" in a and "nr_throttled" in b:
        d["cgroup_periods"] = b["nr_periods"] - a["nr_periods"]
        d["cgroup_periods_throttled"] = b["nr_throttled"] - a["nr_throttled"]
        d["cgroup_throttled_ms"] = round((b["throttled_usec"] - a["throttled_usec"]) / 1e3, 1)
    return d


def proof_algorithmic_bytes(cv, log_n: int, nb_commit: int) -> dict:
    """SURVEY.md section 8d's per-proof algorithmic bytes (the contract's HBM yardstick), per curve and size: 10 KZG commitments of n
    (scalar, point) pairs; the GPU-natural transform schedule 12 iNTT_n + 12 coset-NTT_4n + 1 iNTT_4n at 64 B per element; one
    quotient pass over 14 vectors of 4n elements; ~50 MB of other pointwise passes at 2^17, taken proportional to n.  Each BSB22
    commitment adds one MSM, one iNTT_n + one coset-NTT_4n and two 4n vectors in the quotient pass."""
    n = 1 << log_n
    pair = 32 + 2 * cv.fp_bytes
    msm = (10 + nb_commit) * n * pair
    ntt = ((12 + nb_commit) * n + (13 + nb_commit) * 4 * n) * 64
    quotient = (14 + 2 * nb_commit) * 4 * n * 32
    misc = int(50e6 * n / (1 << 17))
    return {"msm": msm, "ntt": ntt, "quotient": quotient, "misc": misc, "total": msm + ntt + quotient + misc}


def extra_rooflines(args, cv, st, value_per_rank: float, lat_stats_ms: float) -> dict:
    """The whole proof and the transforms on the contract's HBM yardstick, and where a lone instrumented proof's time goes - every
    input measured by this run (apk_stats: HIP events on the kernels' own stream)."""
    alg = proof_algorithmic_bytes(cv, args.log_n, args.bsb22)
    proofs = max(st.proofs, 1)
    out = {"proof": {"bound": "hbm", "algorithmic_bytes": alg["total"], "algorithmic_bytes_by_part": alg,
                     "achieved": round(alg["total"] * value_per_rank / 1e9, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(alg["total"] * value_per_rank / 1e9 / HBM_PEAK_GBS, 6),
                     "basis": "SURVEY.md section 8d bytes per proof x proofs/s of one GPU under load"}}
    if st.ntt_ms > 0:
        gbs = st.ntt_elements * 64.0 / (st.ntt_ms * 1e-3) / 1e9
        out["ntt"] = {"bound": "hbm", "kernel": "ntt_pass_kernel", "algorithmic_bytes_per_element": 64,
                      "elements_per_proof": int(st.ntt_elements / proofs), "ms_per_proof": round(st.ntt_ms / proofs, 4),
                      "achieved": round(gbs, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 6),
                      "basis": "lone instrumented proofs: elements transformed x 64 B (one read + one write per transform) / HIP-event time of the passes"}
    per = lambda ms: round(ms / proofs, 4)
    total = st.prove_ms / proofs if st.prove_ms > 0 else lat_stats_ms
    parts = {"msm_accumulate": per(st.msm_accumulate_ms), "msm_sort_and_scans": per(st.msm_sort_ms), "msm_tails": per(st.msm_tail_ms),
             "ntt": per(st.ntt_ms), "host_lincomb": per(st.host_lincomb_ms)}
    parts["rounds_2_to_4_pointwise_and_host"] = round(max(total - sum(parts.values()), 0.0), 4)
    out["lone_proof_ms_by_part"] = {"total": round(total, 4), **parts,
                                    "share": {k: round(v / total, 4) for k, v in parts.items()} if total > 0 else None,
                                    "basis": "apk_stats over the instrumented lone proofs (the statistics synchronise the host with every batch, so the total is above proof_latency_ms)"}
    return out


def default_witnesses(args) -> int:
    k = args.witnesses or (16 if args.log_n <= 18 else 8 if args.log_n == 19 else 2)
    return max(1, min(k, args.inflight))


def bench_prove(args, cv, rk) -> None:
    from algoplonk_amd import _lib, batch, frontend, plonk, setup, workloads
    from algoplonk_amd._lib import lib, check

    seed = {("bn254", 0): 0xA190, ("bls12_381", 0): 0xA191}.get((args.curve, args.bsb22), 0xA193)
    # ---- the hard input FIRST, on a context of its own that is closed again before the headline context
    # exists (measured behind it, with both contexts alive, the same leg read 7 % low whatever the witness was)
    witness_bits = None
    if rk.world == 1 and args.witness == "uniform" and not args.bsb22 and args.log_n <= 19:
        witness_bits = skewed_leg(args, cv, rk, seed)
    t0 = time.time()
    pi2_host = None
    if args.bsb22:
        ccs, witness, blinding, tau = workloads.random_circuit_bsb22(cv, args.log_n, seed, nb_commitments=args.bsb22)
        name = "%s random circuit, 2^%d constraints, %d BSB22 commitment(s)" % (cv.name, args.log_n, args.bsb22)
    else:
        wl = (workloads.skewed_circuit if args.witness == "bits" else workloads.random_circuit)(cv, args.log_n, seed)
        ccs, witness, blinding, tau, name = wl.ccs, wl.witness, wl.blinding, wl.tau, wl.name
    n = ccs.domain_size()
    srs = setup.unsafe_srs(cv, n, tau, device=rk.local_rank, lagrange=bool(args.bsb22))
    pk, vk = plonk.Setup(ccs, srs, device=rk.local_rank, msm_window=args.msm_window, slots=args.inflight)
    args.window_used = pk.msm_window
    # ---- K distinct assignments of the circuit: the callers of the timed region walk through them, so that at
    # any moment the proofs in flight have different wires, public inputs, blinding scalars (and BSB22 columns / hiding pairs) - as
    # concurrent (*CompiledCircuit).Verify calls do (algoplonk.go:79-98).  Assignment 0 is the workload's own.
    K = default_witnesses(args)
    first = workloads.Variant(witness, blinding, None if args.bsb22 else wl.solution, [(0xA193 + k, 0x3910A + k) for k in range(args.bsb22)])
    vs = [first] + workloads.variant_inputs(ccs, K - 1, seed)
    ws = batch.WitnessSet(pk, ccs, vs).to_device()
    if not args.no_host_inputs:
        ws.to_pinned(rk.local_rank)
    setup_s = time.time() - t0

    proofs = [_lib.Proof() for _ in range(args.inflight)]
    last_pick = [0] * args.inflight
    counters = [0] * args.inflight
    errors = []
    mode = {"where": "device"}
    import math
    stride = next((c for c in (3, 5, 7, 11, 13) if math.gcd(c, K) == 1), 1)

    def one(i):
        a = (i + stride * counters[i]) % K    # caller i walks the set; its neighbours hold other assignments at any moment
        counters[i] += 1
        last_pick[i] = a
        rc = ws.prove(a, proofs[i], mode["where"])
        if rc != 0:
            errors.append((rc, lib.apk_last_error()))

    import hashlib
    from algoplonk_amd import MarshalProof
    sha = lambda p: hashlib.sha256(MarshalProof(plonk.Proof(cv, p))).hexdigest()[:16]

    pk.paths(reset=True)
    cpu0 = host_cpu_snapshot()
    elapsed = rk.timed_callers(one, args.inflight, args.steps, args.warmup, args.step_barrier)
    host_cpu = host_cpu_delta(cpu0, host_cpu_snapshot(), elapsed)       # (includes the callers' warm-up proofs: a few percent)
    if errors:
        raise SystemExit("apk_prove failed: %r" % (errors[0],))
    total_proofs = args.steps * args.inflight * rk.world
    value = total_proofs / elapsed
    # ---- what the timed region PRODUCED: every caller's last proof (made with the other callers in flight, i.e. by the loaded
    # forms of the kernels - `paths` says which ran) is marshalled and hashed before anything else touches the buffers; each is
    # held to the lone proof and to the C oracle's proof of ITS OWN assignment below
    paths_loaded = pk.paths(reset=True)
    load_hashes = [(last_pick[i], sha(proofs[i])) for i in range(args.inflight)]

    # ---- the call the cgo shim makes: apk_prove with HOST pointers, same callers, same walk through the
    # assignments - from page-locked buffers (apk_host_alloc: the shim's witness pool) and from ordinary pageable memory
    host_legs = {}
    if not args.no_host_inputs:
        for where in ("pinned", "pageable"):
            mode["where"] = where
            # (the same number of steps as the headline for the page-locked leg: with persistent callers the ramp-down of the last
            # proofs is a fixed cost, and a shorter leg would read low for that reason alone)
            steps_h = args.steps if where == "pinned" else max(6, args.steps // 2)
            el = rk.timed_callers(one, args.inflight, steps_h, 2, args.step_barrier)
            if errors:
                raise SystemExit("apk_prove (%s host inputs) failed: %r" % (where, errors[0]))
            host_legs[where] = {"value": steps_h * args.inflight * rk.world / el, "steps": steps_h,
                                "hashes": [(last_pick[i], sha(proofs[i])) for i in range(args.inflight)]}
        host_legs["paths_pageable_leg"] = pk.paths(reset=True)
        mode["where"] = "device"

    # ---- single-proof latency + live HIP-event timing of the dominant kernel (own pass, after the timed region)
    # latency as a caller sees it: one proof at a time, instrumentation off (the library's statistics synchronise the host
    # with every MSM and NTT batch, which keeps it from queueing ahead); then the same with the statistics on for the per-round
    # and per-kernel figures below
    nlat = 5
    lone_hashes = []
    for a in range(K):
        check(ws.prove(a, proofs[0], "device"))
        lone_hashes.append(sha(proofs[0]))
    paths_lone = pk.paths(reset=True)

    def lone_latency(where):
        check(ws.prove(0, proofs[0], where))
        t_ = time.perf_counter()
        for k in range(nlat):
            check(ws.prove(k % K, proofs[0], where))
        return (time.perf_counter() - t_) / nlat * 1e3

    lat_ms = lone_latency("device")
    lat_host = {w: lone_latency(w) for w in (("pinned", "pageable") if not args.no_host_inputs else ())}
    pk.enable_stats(True)
    pk.stats(reset=True)
    lat1 = time.perf_counter()
    for k in range(3):
        check(ws.prove(k % K, proofs[0], "device"))
    lat_stats_ms = (time.perf_counter() - lat1) / 3 * 1e3
    st = pk.stats(reset=True)
    pk.enable_stats(False)
    dptr = ws.items[0].dev

    # ---- MSM-only throughput (second half of BASELINE.json's metric): one 2^log_n MSM, scalars resident in HBM
    out_pt = C.create_string_buffer(2 * cv.fp_bytes)
    for _ in range(3):
        check(lib.apk_msm_g1_device(pk.ctx, 0, dptr[0], n, out_pt))
    reps = 20
    m0 = time.perf_counter()
    for _ in range(reps):
        check(lib.apk_msm_g1_device(pk.ctx, 0, dptr[0], n, out_pt))
    msm_s = (time.perf_counter() - m0) / reps
    msm_mscalar = n / msm_s / 1e6
    # the same MSM with the device kept busy: 16 callers (one per proving slot), each issuing its MSMs back to back - the rate
    # the commitments of concurrent proofs actually run at (one MSM at a time leaves the GPU idle through its reduction tail)
    sat_threads, sat_reps = min(16, args.inflight), 12

    def msm_worker():
        o = C.create_string_buffer(2 * cv.fp_bytes)
        for _ in range(sat_reps):
            check(lib.apk_msm_g1_device(pk.ctx, 0, dptr[0], n, o))

    ts = [threading.Thread(target=msm_worker) for _ in range(sat_threads)]
    s0 = time.perf_counter()
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    msm_sat_mscalar = sat_threads * sat_reps * n / (time.perf_counter() - s0) / 1e6

    if witness_bits:
        witness_bits["ratio_to_uniform"] = round(witness_bits["value"] / value, 4)

    pmc = None
    cpu_baseline = None
    oracle_sha = None
    if rk.rank == 0 and rk.world == 1:
        if not args.no_pmc:
            # (the PMC passes profile the MSMs of a circuit of the same size and curve WITHOUT the commitment: the accumulate
            # kernel's traffic and instruction count per pair do not depend on the circuit)
            pmc = pmc_traffic(args.curve, args.log_n, window_bits(args), timeout_s=150.0 if args.log_n < 20 else 1500.0)   # a pass takes ~10 s at 2^17; a hung profiler must not hold the line up
        if not args.no_oracle_check:
            # the CHECKER: the host prover's proof of every assignment (oracle/fast_prover.c, ~0.4 s each at 2^17 on 16 cores)
            try:
                from bench_cpu import oracle_blobs
                oracle_sha = [hashlib.sha256(b).hexdigest()[:16] if b else None for b in oracle_blobs(cv, ccs, srs, ws.items)]
            except Exception as e:
                oracle_sha = {"error": str(e)[:200]}
        if not args.no_cpu_baseline:
            probe = go_probe()
            cpu_baseline = gnark_cpu_baseline(probe, args.curve, args.log_n, args.cpu_baseline_seconds)
            if cpu_baseline is None:
                try:
                    from bench_cpu import cpu_baseline_prove_items
                    cpu_baseline = cpu_baseline_prove_items(cv, name, ccs, srs, ws.items[0], args.cpu_baseline_seconds)
                except Exception as e:  # the baseline is reported, never required for the GPU number
                    cpu_baseline = {"value": None, "unit": "proofs/sec", "cores": 0, "kind": "port", "sample": "unavailable: %s" % e}
            cpu_baseline["go_probe"] = probe
    roofline = roofline_from_stats(args, cv, st, pmc, valu_issue_rate() if (rk.rank == 0 and pmc) else None)
    roofline.update(extra_rooflines(args, cv, st, value / rk.world, lat_stats_ms))
    # the whole device under load against its VALU issue rate: instructions of a proof made under load (counter runs of this
    # invocation) x proofs/s, priced at the issue rate of the accumulate kernel's instruction mix (the bulk of them)
    v1 = roofline.get("valu") or {}
    if rk.rank == 0 and rk.world == 1 and pmc and v1.get("ns_per_wave_instruction_per_simd"):
        vp = pmc_valu_per_proof(args.curve, args.log_n, args.inflight, args.bsb22, timeout_s=300.0 if args.log_n < 20 else 2400.0)
        if vp:
            bound = v1["simds"] / (vp["wave_instructions_per_proof"] * v1["ns_per_wave_instruction_per_simd"] * 1e-9)
            roofline["valu_under_load"] = {"wave_instructions_per_proof": round(vp["wave_instructions_per_proof"], 1),
                                           "ns_per_wave_instruction_per_simd": v1["ns_per_wave_instruction_per_simd"], "simds": v1["simds"],
                                           "issue_bound_proofs_per_s": round(bound, 2), "proofs_per_s": round(value, 2), "frac": round(value / bound, 4),
                                           "basis": "SQ_INSTS_VALU over every kernel of %d proofs made by %d concurrent callers (two counter runs of this invocation, "
                                                    "their difference) x this run's proofs/s, at the issue rate of the accumulate kernel's instruction mix" % (vp["proofs_between_the_runs"], vp["callers"])}
    plane = rk.data_plane_probe(pk.ctx)

    # ---- parity of everything this run produced.  Per assignment a: the lone proof's hash; every proof made under load for a (the
    # callers' last proofs of the timed region and of the host-input legs) must equal it; and it must equal the C oracle's proof
    # of a.  Every rank checks its own region; the line (rank 0) carries the verdict of all of them.
    def leg_ok(hs):
        return all(h == lone_hashes[a] for a, h in hs)

    loaded_ok = leg_ok(load_hashes)
    host_ok = all(leg_ok(host_legs[w]["hashes"]) for w in ("pinned", "pageable") if w in host_legs)
    all_ok = rk.comm.max(0.0 if (loaded_ok and host_ok) else 1.0) == 0.0
    matches_oracle = None
    if isinstance(oracle_sha, list):
        matches_oracle = [oracle_sha[a] == lone_hashes[a] for a in range(K)]
    if rk.rank == 0:
        if cpu_baseline and cpu_baseline.get("proof_sha256_prefix"):
            cpu_baseline["matches_gpu_proof"] = cpu_baseline["proof_sha256_prefix"] == lone_hashes[0]
            cpu_baseline["matches_proofs_under_load"] = loaded_ok and cpu_baseline["proof_sha256_prefix"] == lone_hashes[0]
        line = {
            "metric": "proofs/sec", "value": round(value, 4), "unit": "proofs/sec", "n_gpus": rk.world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u32x8 (Montgomery Fr/Fp)" if cv.name == "bn254" else "u32x8 Fr / u32x12 Fp (Montgomery)",
            "data": "synthetic",
            "config": {"workload": name, "log_n": args.log_n, "curve": cv.name, "proofs_per_step": args.inflight,
                       "distinct_witnesses": K,
                       "stepping": "joined after every step" if (args.step_barrier or args.inflight == 1) else
                                   "%d persistent callers x K proofs each, every caller walking through the %d assignments; barriers only around the K steps" % (args.inflight, K),
                       "inputs": "L, R, O resident in HBM (apk_prove_device); the host-pointer call of the cgo shim is `value_host_inputs`",
                       "srs": "synthetic tau=SHA256(seed)", "parallelism": "replicas x%d" % rk.world, "world_size": rk.world,
                       "backend": "libapk comm (barrier + MAX over TCP only: independent proofs exchange no data)" if rk.world > 1 else "single process",
                       "rccl_ranks": plane.get("rccl_ranks", 0), "data_plane": plane},
            # the timed region's own output, per assignment
            "distinct_witnesses": K, "matches_oracle": matches_oracle,
            "witness_sha256_prefixes": lone_hashes, "oracle_sha256_prefixes": oracle_sha,
            "proofs_under_load_checked": len(load_hashes), "proofs_under_load_match_lone_proofs": loaded_ok,
            "proofs_under_load_ok_on_all_ranks": all_ok,
            # apk_prove with host pointers: the same callers and assignments from page-locked / pageable host memory
            "value_host_inputs": round(host_legs["pinned"]["value"], 4) if "pinned" in host_legs else None,
            "value_host_inputs_pageable": round(host_legs["pageable"]["value"], 4) if "pageable" in host_legs else None,
            "host_inputs_ratio": round(host_legs["pinned"]["value"] / value, 4) if "pinned" in host_legs else None,
            "host_input_steps": {w: host_legs[w]["steps"] for w in ("pinned", "pageable") if w in host_legs},
            "proofs_from_host_inputs_match_lone_proofs": host_ok if host_legs else None,
            "proof_latency_host_inputs_ms": round(lat_host["pinned"], 3) if lat_host else None,
            "proof_latency_host_inputs_pageable_ms": round(lat_host["pageable"], 3) if lat_host else None,
            "proof_latency_ms": round(lat_ms, 3), "proof_latency_instrumented_ms": round(lat_stats_ms, 3), "msm_mscalar_per_s": round(msm_mscalar, 3),
            "msm_ms": round(msm_s * 1e3, 4), "msm_mscalar_per_s_saturated": round(msm_sat_mscalar, 3), "setup_s": round(setup_s, 2),
            "msm_batch_avg_ms": round(st.msm_total_ms / max(st.msm_batches, 1), 4),
            "ntt_ms_per_proof": round(st.ntt_ms / max(st.proofs, 1), 4),
            "round_ms": [round(x / max(st.proofs, 1), 3) for x in st.round_ms],
            "host_lincomb_ms": round(st.host_lincomb_ms / max(st.proofs, 1), 3),
            "witness_bits": witness_bits,
            # every run-time knob that was set for this run (none = the library's defaults): a line is reproducible from its command
            # line plus this
            "env_knobs": {k: v for k, v in sorted(os.environ.items()) if k.startswith("APK_") and k != "APK_COMM_TOKEN"},
            "msm_window": pk.msm_window,
            "paths_under_load": paths_loaded, "paths_lone_proof": paths_lone,
            "host_cpu_timed_region": host_cpu,
            "proof_sha256_prefix": lone_hashes[0], "roofline": roofline, "cpu_baseline": cpu_baseline,
        }
        print(json.dumps(line), flush=True)
    ws.close()
    rk.close()
    bad = []
    if not all_ok:
        wrong = sorted({(a, h) for a, h in load_hashes if h != lone_hashes[a]})
        bad.append("proofs of the timed region%s differ from the lone proofs of the same assignments (rank %d: %s)"
                   % ("" if not loaded_ok else " (host-input legs)", rk.rank, wrong[:4]))
    if rk.rank == 0 and matches_oracle is not None and not all(matches_oracle):
        bad.append("proofs differ from the C oracle's proofs of the same assignments: %s" % [a for a, m in enumerate(matches_oracle) if not m])
    if bad:
        raise SystemExit("bench.py: PARITY FAILURE - " + "; ".join(bad))


def skewed_leg(args, cv, rk, seed):
    """proofs/s on workloads.skewed_circuit (~80 % of the wire values in {0, 1}, the rest 16-bit or uniform), same size, same
    number of concurrent callers, a third of the steps.  This prover commits the CANONICAL (blinded) wire polynomials: after the
    iNTT the MSM scalars are uniform whatever the witness looks like, so the rate must match the headline - the line checks it."""
    from algoplonk_amd import _lib, frontend, plonk, setup, workloads
    from algoplonk_amd._lib import lib, check
    wl = workloads.skewed_circuit(cv, args.log_n, seed ^ 0x5EED)
    srs = setup.unsafe_srs(cv, wl.ccs.domain_size(), wl.tau, device=rk.local_rank)
    pk, vk = plonk.Setup(wl.ccs, srs, device=rk.local_rank, msm_window=args.msm_window, slots=args.inflight)
    L, R, O = frontend.wire_columns(wl.ccs, wl.solution)
    small = sum(1 for col in (L, R, O) for v in col if v < 2) / (3.0 * len(L))
    dptr = []
    for vec in (L, R, O):
        b = cv.fr_vector(vec)
        p = C.c_void_p()
        check(lib.apk_device_alloc(pk.ctx, len(b), C.byref(p)))
        check(lib.apk_device_upload(pk.ctx, p, b, len(b)))
        dptr.append(p)
    pub, bl = cv.fr_vector(wl.witness.public), cv.fr_vector(wl.blinding)
    proofs = [_lib.Proof() for _ in range(args.inflight)]
    errors = []

    def one(i):
        rc = lib.apk_prove_device(pk.ctx, dptr[0], dptr[1], dptr[2], pub, bl, None, C.byref(proofs[i]))
        if rc != 0:
            errors.append((rc, lib.apk_last_error()))

    steps = max(6, args.steps // 3)
    elapsed = rk.timed_callers(one, args.inflight, steps, args.warmup, args.step_barrier)
    ok = not errors
    if ok:
        try:
            plonk.Verify(plonk.Proof(cv, proofs[0]), _with_g2(vk, cv, wl.tau), wl.witness)
        except Exception:
            ok = False
    pk.close()
    value = steps * args.inflight / elapsed
    return {"workload": wl.name, "wire_values_in_0_1": round(small, 3), "value": round(value, 3), "unit": "proofs/sec", "steps": steps,
            "ratio_to_uniform": None, "proof_verifies": ok}


def _with_g2(vk, cv, tau):
    from algoplonk_amd import setup
    vk.KzgG2 = setup.g2_from_tau(cv, tau)
    return vk


def bench_sharded_msm(args, cv, rk) -> None:
    """BASELINE.json configs[3]: one 2^log_n MSM (seed 0xA192, uniform scalars, SRS-shaped points) sharded by index range:
    every rank keeps the windowed tables of its slice resident, computes a full partial sum, then ONE all-gather of a
    64/96-byte point per rank + world-1 host point additions (apk_msm_g1_sharded, csrc/comm.cpp).  Strong scaling."""
    from algoplonk_amd import parallel, plonk, setup, workloads
    from algoplonk_amd import _lib
    from algoplonk_amd._lib import lib, check

    n = 1 << args.log_n
    g = workloads.SplitMix64(0xA192)
    tau = workloads.tau_from_seed(0xA192, cv.r)
    srs = setup.unsafe_srs(cv, n, tau, device=rk.local_rank)
    scalars = cv.fr_vector([g.fr(cv.r) for _ in range(n)])
    bases = srs.g1[: n * 2 * cv.fp_bytes]
    sm = parallel.ShardedMsm(cv, bases, device=rk.local_rank, comm=rk.comm, msm_window=args.msm_window)
    sm.upload(scalars)                            # this rank's slice of the scalars resident in HBM
    args.window_used = lib.apk_ctx_msm_window(sm._ctx)
    plane = rk.plane_report()
    d = sm._d
    out = C.create_string_buffer(2 * cv.fp_bytes)
    res = [b""]

    def step():
        res[0] = sm.run()                         # apk_msm_g1_sharded: local MSM, ONE all-gather of a point per rank, local additions

    elapsed = rk.timed(step, args.steps, args.warmup)
    # dominant kernel of this rank's share, HIP events on its stream
    check(lib.apk_stats_enable(sm._ctx, 1))
    st = _lib.Stats()
    check(lib.apk_stats_read(sm._ctx, C.byref(st), 1))
    for _ in range(5):
        check(lib.apk_msm_g1_device(sm._ctx, 0, d, sm.hi - sm.lo, out))
    check(lib.apk_stats_read(sm._ctx, C.byref(st), 1))
    check(lib.apk_stats_enable(sm._ctx, 0))
    pmc = None
    cpu_baseline = None
    if rk.rank == 0 and rk.world == 1:
        if not args.no_pmc:
            pmc = pmc_traffic(args.curve, args.log_n, window_bits(args))
        if not args.no_cpu_baseline:
            from bench_cpu import cpu_baseline_msm
            cpu_baseline = cpu_baseline_msm(cv, bases, scalars, n, args.cpu_baseline_seconds)
            cpu_baseline["go_probe"] = go_probe()
            cpu_baseline["matches_gpu_result"] = cpu_baseline.pop("result") == res[0]
    if rk.rank == 0:
        import hashlib
        print(json.dumps({
            "metric": "MSM Mscalar/s", "value": round(n * args.steps / elapsed / 1e6, 3), "unit": "Mscalar/s", "n_gpus": rk.world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "u29x9 Fp (BN254) / u28x14 Fp (BLS12-381) unsaturated Montgomery",
            "data": "synthetic", "config": {"workload": "%s single MSM 2^%d sharded by index range" % (cv.name, args.log_n),
                                            "parallelism": "index-range x%d + all-gather of %d-byte points" % (rk.world, 2 * cv.fp_bytes),
                                            "world_size": rk.world, "backend": ("libapk comm: tcp control plane, %s data plane" % rk.comm.transport) if rk.world > 1 else "single process",
                                            "rccl_ranks": rk.comm.rccl_ranks, "data_plane": plane},
            "result_sha256_prefix": hashlib.sha256(res[0]).hexdigest()[:16],
            "roofline": roofline_from_stats(args, cv, st, pmc, valu_issue_rate() if pmc else None), "cpu_baseline": cpu_baseline}), flush=True)
    rk.close()


if __name__ == "__main__":
    main()
