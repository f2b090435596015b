import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from algoplonk_amd import ecc, parallel, plonk, setup, workloads, MarshalProof
from algoplonk_amd._lib import lib, check
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 9
cv = ecc.BN254 if (len(sys.argv) < 3 or sys.argv[2] == "bn254") else ecc.BLS12_381
wl = workloads.random_circuit(cv, log_n, 0xA190)
srs = setup.unsafe_srs(cv, wl.ccs.domain_size(), wl.tau)
pk, vk = plonk.Setup(wl.ccs, srs)
plain = MarshalProof(plonk.Prove(wl.ccs, pk, wl.witness, wl.blinding))
print("plain ok", flush=True)
import torch
print("torch.cuda.is_available:", torch.cuda.is_available(), flush=True)
sc = parallel.SplitCommitter(cv, pk.ctx, 0, 1)
print("dev", sc.dev, flush=True)
orig = sc.commit
def commit(basis, d, lens):
    print("commit", basis, [hex(x) for x in d], lens, flush=True)
    r = orig(basis, d, lens)
    print(" ->", [x[:8].hex() for x in r], flush=True)
    return r
sc.commit = commit
sc.install()
hooked = MarshalProof(plonk.Prove(wl.ccs, pk, wl.witness, wl.blinding))
print("hooked == plain:", hooked == plain, sc.batches, flush=True)
sc.stop()
