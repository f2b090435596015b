#!/usr/bin/env python3
"""Driver for counter runs over the LOADED prover: `callers` threads prove `rounds` proofs each (distinct witnesses) on one context,
so the kernels take the forms they take under load (lean tails, long units, radix-4 steps, gangs).  Under rocprofv3 --pmc the
dispatches are serialised by the profiler, but the callers are still inside apk_prove together - which is what the forms follow.
usage: python tools/prof_loaded_proofs.py <bn254|bls12_381> <log_n> <callers> <rounds> [bsb22]
Two runs with different `rounds` and the same everything else differ by exactly (rounds2 - rounds1) x callers proofs: their counter
totals' difference is the cost of those proofs, whatever the context's creation launched (bench.py valu_under_load)."""
import os
import sys
import threading

os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from algoplonk_amd import _lib, batch, ecc, plonk, setup, workloads

cv = ecc.BLS12_381 if sys.argv[1] == "bls12_381" else ecc.BN254
log_n, callers, rounds = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
bsb = int(sys.argv[5]) if len(sys.argv) > 5 else 0
seed = 0xA193 if bsb else (0xA190 if cv is ecc.BN254 else 0xA191)
if bsb:
    ccs, w, bl, tau = workloads.random_circuit_bsb22(cv, log_n, seed, nb_commitments=bsb)
    first = workloads.Variant(w, bl, None, [(0xA193 + k, 0x3910A + k) for k in range(bsb)])
else:
    wl = workloads.random_circuit(cv, log_n, seed)
    ccs, tau = wl.ccs, wl.tau
    first = workloads.Variant(wl.witness, wl.blinding, wl.solution, [])
srs = setup.unsafe_srs(cv, ccs.domain_size(), tau, lagrange=bool(bsb))
pk, vk = plonk.Setup(ccs, srs, slots=callers)
K = 1 if log_n >= 20 else min(4, callers)       # (assignments of a 2^21 circuit take minutes to generate in Python)
ws = batch.WitnessSet(pk, ccs, [first] + workloads.variant_inputs(ccs, K - 1, seed)).to_device()
bad = []


def worker(i):
    pr = _lib.Proof()
    for r in range(rounds):
        if ws.prove((i + r) % K, pr, "device") != 0:
            bad.append(_lib.lib.apk_last_error())
            return


th = [threading.Thread(target=worker, args=(i,)) for i in range(callers)]
[t.start() for t in th]
[t.join() for t in th]
print("proofs %d errors %d paths %s" % (callers * rounds, len(bad), pk.paths()))
ws.close()
sys.exit(1 if bad else 0)
