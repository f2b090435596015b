#!/usr/bin/env python3
"""Regenerates tests/golden/* .  Run in the build container (needs /root/reference); the outputs are data only.

1. Data files the reference's own tests hold (setup/trusted_setup_test.go reads them through setup.go):
   - the three vk.bin files (160 / 240 / 240 bytes)
   - the head of the Ethereum KZG ceremony pk.bin (count + first 8 compressed G1 points) and point 32767
   plus the hex known answers asserted at setup/trusted_setup_test.go:53-59,132,183-189,256, as JSON.
2. Golden proof vectors produced by the Python oracle (oracle/plonk.py) on seeded circuits: every input is
   re-derivable from (curve, log_n, circuit seed, tau seed, blinding seed); the JSON stores the expected proof blob,
   public-input blob and challenges.  The C oracle (CPU tier) and the HIP path (GPU tier) must reproduce them.
"""
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = "/root/reference"


def trusted_setup_fixtures():
    for name in ("PerpetualPowersOfTauBN254", "EethereumKzgCeremonyBLS12_381", "DuskBLS12_381"):
        with open(os.path.join(REF, "setup", name, "vk.bin"), "rb") as f, open(os.path.join(HERE, name + ".vk.bin"), "wb") as o:
            o.write(f.read())
    with open(os.path.join(REF, "setup", "EethereumKzgCeremonyBLS12_381", "pk.bin"), "rb") as f:
        pk = f.read()
    with open(os.path.join(HERE, "EethereumKzgCeremonyBLS12_381.pk.head.bin"), "wb") as o:
        o.write(pk[: 4 + 8 * 48])
    with open(os.path.join(HERE, "EethereumKzgCeremonyBLS12_381.pk.32767.bin"), "wb") as o:
        o.write(pk[4 + 32767 * 48: 4 + 32768 * 48])
    # known answers = the hex literals of the reference's test file
    src = open(os.path.join(REF, "setup", "trusted_setup_test.go")).read()
    dusk, eth = src.split("func TestTrustedSetupEethereumKzgCeremonyBLS12_381")
    g1 = lambda s: re.findall(r'"0x([0-9a-f]{96})"', s)
    kat = {"source": "setup/trusted_setup_test.go", "dusk_g1_first5": g1(dusk)[:5], "dusk_g1_32767": g1(dusk)[5],
           "ethereum_g1_first5": g1(eth)[:5], "ethereum_g1_32767": g1(eth)[5]}
    json.dump(kat, open(os.path.join(HERE, "trusted_setup_kat.json"), "w"), indent=1)


def proof_vectors():
    from algoplonk_amd import ecc
    from helpers import CURVES, blinding, oracle_circuit_from_ccs, random_chain_ccs
    from oracle import plonk as oplonk
    from oracle.prng import tau_from_seed

    vecs = []
    for cname in ("bn254", "bls12-381"):
        cv, ov = CURVES[cname]
        for log_n, cseed, tseed, bseed in ((3, 0xA190, 1, 7), (5, 0xA191, 2, 8), (8, 0xA192, 3, 9), (10, 0xA193, 4, 10)):
            ccs, w, sol = random_chain_ccs(cv, log_n, cseed)
            n = ccs.domain_size()
            tau = tau_from_seed(tseed, cv.r)
            opk = oplonk.setup(oracle_circuit_from_ccs(ov, ccs), oplonk.synthetic_srs(ov, n, tau, materialize=False))
            oc = oracle_circuit_from_ccs(ov, ccs)
            L, R, O = oplonk.solve_lro(oc, sol)
            bl = blinding(cv, bseed)
            tr = oplonk.ProverTrace()
            pr = oplonk.prove(opk, L, R, O, w.public, bl, trace_out=tr)
            blob = oplonk.marshal_proof(ov, pr)
            pib = oplonk.marshal_public_inputs(w.public)
            assert oplonk.verify(opk.vk, blob, pib)
            vecs.append({"curve": cname, "log_n": log_n, "circuit_seed": cseed, "tau_seed": tseed, "blinding_seed": bseed,
                         "proof": blob.hex(), "public_inputs": pib.hex(),
                         "challenges": {k: hex(getattr(tr, k)) for k in ("gamma", "beta", "alpha", "zeta", "gamma_kzg")},
                         "vk": {"ql": ov.raw_bytes(opk.vk.ql).hex(), "s3": ov.raw_bytes(opk.vk.s[2]).hex()}})
            print("vector", cname, log_n, len(blob))
    json.dump({"generator": "tests/golden/make_fixtures.py (oracle/plonk.py)", "vectors": vecs},
              open(os.path.join(HERE, "proof_vectors.json"), "w"), indent=1)




def ethereum_srs_prefix():
    """First 2^14 + 3 points of the Ethereum KZG ceremony pk.bin (what setup.Run loads for a 2^14 circuit,
    setup/setup.go:113-114,196-228) + its vk.bin, laid out like the reference's setup/<NamePath>/ directory."""
    d = os.path.join(HERE, "setup", "EethereumKzgCeremonyBLS12_381")
    os.makedirs(d, exist_ok=True)
    src = os.path.join(REF, "setup", "EethereumKzgCeremonyBLS12_381")
    with open(os.path.join(src, "pk.bin"), "rb") as f, open(os.path.join(d, "pk.bin"), "wb") as o:
        o.write(f.read(4 + 16387 * 48))
    with open(os.path.join(src, "vk.bin"), "rb") as f, open(os.path.join(d, "vk.bin"), "wb") as o:
        o.write(f.read())


if __name__ == "__main__":
    trusted_setup_fixtures()
    ethereum_srs_prefix()
    proof_vectors()
