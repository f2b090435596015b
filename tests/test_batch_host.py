"""CPU tier: the distinct-assignment plumbing of the under-load parity checks, without a GPU - workloads.variants gives distinct
satisfying assignments of one circuit, batch.WitnessSet packs them into gnark's layout, and bench_cpu.oracle_blobs (the checker of
bench.py's timed region, tools/soak.py and tests/test_gpu_load.py) returns, per assignment, the bytes of the Python oracle's proof."""
import ctypes as C

import pytest

from algoplonk_amd import _lib, batch, frontend, setup as ap_setup, workloads
from algoplonk_amd._lib import lib
from bench_cpu import oracle_blobs
from oracle import plonk as oplonk

from helpers import CURVES, oracle_circuit_from_ccs


@pytest.mark.parametrize("cname", ["bn254", "bls12-381"])
def test_oracle_blobs_per_assignment(cname):
    cv, ov = CURVES[cname]
    wl = workloads.random_circuit(cv, 6, 0x51DE)
    n = wl.ccs.domain_size()
    vs = workloads.variants(wl, 4, 0x51DE)
    assert vs[0].solution is wl.solution and len({tuple(v.witness.Vector()) for v in vs}) == 4 and len({tuple(v.blinding) for v in vs}) == 4
    for v in vs:                       # every variant satisfies every gate
        for ql, qr, qm, qo, qk, xa, xb, xc in wl.ccs.constraints:
            s = v.solution
            assert (ql * s[xa] + qr * s[xb] + qm * s[xa] * s[xb] + qo * s[xc] + qk) % cv.r == 0
    ws = batch.WitnessSet(None, wl.ccs, vs, curve=cv)
    srs = ap_setup.SRS(cv, n, cv.g1_vector([ov.mul(ov.g1, pow(wl.tau, i, cv.r)) for i in range(n + 3)]), None, wl.tau, None)
    got = oracle_blobs(cv, wl.ccs, srs, ws.items, threads=2, check_first_against_plain=True)
    oc = oracle_circuit_from_ccs(ov, wl.ccs)
    opk = oplonk.setup(oc, oplonk.synthetic_srs(ov, n, wl.tau, materialize=False))
    for v, blob in zip(vs, got):
        L, R, O = oplonk.solve_lro(oc, v.solution)
        assert blob == oplonk.marshal_proof(ov, oplonk.prove(opk, L, R, O, v.witness.public, v.blinding))
    assert len(set(got)) == 4
    before = ws.items[2].O
    ws.corrupt(2)
    assert ws.items[2].O != before and len(ws.items[2].O) == len(before)


def test_host_memory_entry_points_need_a_gpu():
    if _lib.device_count() > 0:
        pytest.skip("a GPU is present")
    p = C.c_void_p()
    assert lib.apk_host_alloc(0, 4096, C.byref(p)) == _lib.APK_ERR_HIP and not p.value
    buf = C.create_string_buffer(4096)
    assert lib.apk_host_register(buf, 4096) == _lib.APK_ERR_HIP
    assert lib.apk_host_alloc(0, 4096, None) == _lib.APK_ERR_ARG
    assert lib.apk_host_free(None) == 0 and lib.apk_host_unregister(None) == 0
