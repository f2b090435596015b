#!/usr/bin/env python3
"""bench.py's cpu_baseline leg on its own (the host provers of oracle/ on this box's cores; the SRS still comes from the GPU).
usage: python tools/cpu_baseline_only.py [bn254|bls12_381] [log_n] [seconds]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from algoplonk_amd import ecc, setup, workloads
from bench_cpu import cpu_baseline_prove

cv = ecc.BLS12_381 if len(sys.argv) > 1 and sys.argv[1] == "bls12_381" else ecc.BN254
log_n = int(sys.argv[2]) if len(sys.argv) > 2 else 17
secs = float(sys.argv[3]) if len(sys.argv) > 3 else 20.0
wl = workloads.random_circuit(cv, log_n, 0xA190 if cv is ecc.BN254 else 0xA191)
srs = setup.unsafe_srs(cv, wl.ccs.domain_size(), wl.tau)
print(json.dumps(cpu_baseline_prove(wl, srs, secs)))
