"""Shared plumbing for the parity tests: build the same circuit for the oracle and for the HIP path."""
from __future__ import annotations

from typing import List, Tuple

from algoplonk_amd import ecc, frontend, plonk as ap_plonk, setup as ap_setup
from oracle import curves as ocurves, plonk as oplonk
from oracle.prng import SplitMix64

def oracle_threads() -> int:
    """pthreads for the C oracle: the CPUs this process may really use (the GPU boxes show 256 CPUs behind a cgroup quota of 16;
    256 threads there are throttled time slices, not parallelism)."""
    import os
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except Exception:
        pass
    return max(1, n)


CURVES = {"bn254": (ecc.BN254, ocurves.BN254), "bls12-381": (ecc.BLS12_381, ocurves.BLS12_381)}


def oracle_circuit_from_ccs(ocv, ccs: frontend.ConstraintSystem) -> oplonk.Circuit:
    return oplonk.Circuit(ocv, ccs.GetNbPublicVariables(), ccs.nb_variables, list(ccs.constraints))


def blinding(cv, seed: int) -> List[int]:
    g = SplitMix64(seed)
    return [g.fr(cv.r) for _ in range(9)]


def oracle_vk_from_product(ovk_curve, vk: ap_plonk.VerifyingKey) -> oplonk.VerifyingKey:
    """Feed the verifying key the HIP path produced to the oracle's transcribed verifier."""
    return oplonk.VerifyingKey(
        curve=ovk_curve, size=vk.Size, size_inv=vk.SizeInv, generator=vk.Generator, coset_shift=vk.CosetShift,
        nb_public=vk.NbPublicVariables, ql=vk.Ql, qr=vk.Qr, qm=vk.Qm, qo=vk.Qo, qk=vk.Qk, s=list(vk.S), qcp=list(vk.Qcp),
        commitment_constraint_indexes=list(vk.CommitmentConstraintIndexes), g1=vk.KzgG1, tau=vk.tau)


def random_chain_ccs(cv, log_n: int, seed: int, nb_public: int = 2) -> Tuple[frontend.ConstraintSystem, frontend.Witness, List[int]]:
    """BASELINE.md §2 random circuit, built directly as a ConstraintSystem (no per-gate closures)."""
    r = cv.r
    n = 1 << log_n
    g = SplitMix64(seed)
    sol = [g.fr(r) for _ in range(nb_public + 2)]
    cons = []
    for _ in range(n - nb_public):
        nv = len(sol)
        xa, xb = g.below(nv), g.below(nv)
        ql, qr, qm, qk = g.fr(r), g.fr(r), g.fr(r), g.fr(r)
        a, b = sol[xa], sol[xb]
        cons.append((ql, qr, qm, r - 1, qk, xa, xb, nv))
        sol.append((ql * a + qr * b + qm * a % r * b + qk) % r)
    ccs = frontend.ConstraintSystem(r, ["p%d" % i for i in range(nb_public)], ["s0", "s1"], cons, "gates", len(sol))
    w = frontend.Witness(r, sol[:nb_public], sol[nb_public:nb_public + 2])
    return ccs, w, sol


def check_bench_line(d: dict, tol: float = 0.01) -> None:
    """Recompute every derived figure of a bench.py line from the line's OWN inputs (bench.py roofline_from_stats):
    value from steps / proofs per step / ms_per_step; roofline.achieved = algorithmic bytes per launch / launch time;
    frac = achieved / peak; hbm_traffic_frac = traffic / launch time / peak; valu.bucket_additions_per_s = pairs x windows /
    launch time; valu.frac = additions / issue_bound.  Used on live output (GPU tier) and on the committed lines (CPU tier)."""
    def close(a, b, what):
        assert abs(a - b) <= tol * max(abs(a), abs(b), 1e-12), (what, a, b)

    if d.get("metric") == "proofs/sec" and "proofs_per_step" in d.get("config", {}):
        close(d["value"], d["steps"] * d["config"]["proofs_per_step"] * d["n_gpus"] / (d["ms_per_step"] * d["steps"] / 1e3), "value")
    rf = d["roofline"]
    assert rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    t = rf["avg_launch_ms"] * 1e-3
    close(rf["achieved"], rf["pairs_per_launch"] * rf["algorithmic_bytes_per_pair"] / t / 1e9, "roofline.achieved")
    close(rf["frac"], rf["achieved"] / rf["peak"], "roofline.frac")
    if rf.get("traffic") is not None and "hbm_traffic_frac" in rf:
        close(rf["hbm_traffic_frac"], rf["traffic"] / t / 1e9 / rf["peak"], "roofline.hbm_traffic_frac")
    v = rf.get("valu")
    if v:
        close(v["bucket_additions_per_s"], rf["pairs_per_launch"] * v["windows"] / t / 1e9, "valu.bucket_additions_per_s")
        if "issue_bound" in v:
            close(v["frac"], v["bucket_additions_per_s"] / v["issue_bound"], "valu.frac")
            # the bound prices this run's instruction count at rates a micro-benchmark measured on the same box: an estimate of the
            # issue limit, good to a few percent - a rate far above it means its basis is stale
            # (every committed line since round 3 reads 0.56 .. 1.001; round 5's 1.12 was slack for a bound priced from constants,
            # which round 5 itself replaced by in-run measurements)
            assert v["frac"] <= 1.05, "an addition rate above the kernel's own issue bound means the bound's basis is stale"
            if "instructions_per_addition" in v:      # round 5: both factors of the bound are on the line
                close(v["issue_bound"], v["simds"] * 64 / (v["instructions_per_addition"] * v["ns_per_wave_instruction_per_simd"]), "valu.issue_bound")
    # round 6: the whole proof and the transforms on the same yardstick, and the parts of a lone proof
    pr = rf.get("proof")
    if pr:
        parts = pr["algorithmic_bytes_by_part"]
        assert pr["algorithmic_bytes"] == parts["total"] == parts["msm"] + parts["ntt"] + parts["quotient"] + parts["misc"]
        log_n, fpb = d["config"]["log_n"], (32 if d["config"]["curve"] == "bn254" else 48)
        n = 1 << log_n
        k = (parts["msm"] // (n * (32 + 2 * fpb))) - 10                     # BSB22 commitments
        assert 0 <= k <= 2 and parts["msm"] == (10 + k) * n * (32 + 2 * fpb)
        assert parts["ntt"] == ((12 + k) * n + (13 + k) * 4 * n) * 64 and parts["quotient"] == (14 + 2 * k) * 4 * n * 32
        close(pr["achieved"], pr["algorithmic_bytes"] * d["value"] / d["n_gpus"] / 1e9, "roofline.proof.achieved")
        close(pr["frac"], pr["achieved"] / pr["peak"], "roofline.proof.frac")
        assert pr["frac"] <= 1.0
    nt = rf.get("ntt")
    if nt:
        close(nt["achieved"], nt["elements_per_proof"] * nt["algorithmic_bytes_per_element"] / (nt["ms_per_proof"] * 1e-3) / 1e9, "roofline.ntt.achieved")
        close(nt["frac"], nt["achieved"] / nt["peak"], "roofline.ntt.frac")
        assert abs(nt["ms_per_proof"] - d["ntt_ms_per_proof"]) < 1e-3
    vl = rf.get("valu_under_load")
    if vl:
        close(vl["issue_bound_proofs_per_s"], vl["simds"] / (vl["wave_instructions_per_proof"] * vl["ns_per_wave_instruction_per_simd"] * 1e-9), "valu_under_load.bound")
        close(vl["frac"], d["value"] / vl["issue_bound_proofs_per_s"], "valu_under_load.frac")
        assert vl["frac"] <= 1.05
    lp = rf.get("lone_proof_ms_by_part")
    if lp and lp.get("share"):
        tot = lp["total"]
        s_parts = sum(lp[k] for k in lp["share"])
        assert s_parts <= tot * 1.02 + 1e-6, (s_parts, tot)
    if "distinct_witnesses" in d:
        assert d["distinct_witnesses"] == d["config"]["distinct_witnesses"] == len(d["witness_sha256_prefixes"])
        assert len(set(d["witness_sha256_prefixes"])) == d["distinct_witnesses"], "the assignments are meant to give distinct proofs"
        if d.get("matches_oracle") is not None:
            assert len(d["matches_oracle"]) == d["distinct_witnesses"] and all(d["matches_oracle"])
        if d.get("value_host_inputs") is not None:
            close(d["host_inputs_ratio"], d["value_host_inputs"] / d["value"], "host_inputs_ratio")
