"""ctypes binding of libapk.so (include/apk.h) - the same C-ABI a cgo shim would bind.

The library is built in-tree by `make -C algoplonk_amd/csrc` (or `__graft_entry__.build()`).  There is no
fallback: if the shared object is missing, importing this module raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# APK_LIB: another build of the same ABI (A/B measurements: tools/ab_bench.sh); default = the in-tree build
LIB_PATH = os.environ.get("APK_LIB") or os.path.join(_HERE, "libapk.so")

APK_BN254 = 0
APK_BLS12_381 = 1
APK_OK = 0
APK_ERR_ARG, APK_ERR_HIP, APK_ERR_STATE, APK_ERR_WITNESS, APK_ERR_VERIFY = 1, 2, 3, 4, 5
ABI_VERSION = 5
G1_MAX = 96
G2_MAX = 192
MAX_COMMITMENTS = 2
NB_BLINDING = 9

# every symbol include/apk.h declares (tests/test_abi.py checks the header against this list)
SYMBOLS = [
    "apk_last_error", "apk_abi_version", "apk_device_count", "apk_g1_bytes", "apk_fp_bytes",
    "apk_ctx_create", "apk_ctx_destroy", "apk_msm_ctx_create", "apk_ctx_get_vk", "apk_msm_g1", "apk_msm_g1_device", "apk_msm_g1_batch_device", "apk_ctx_set_commit_hook", "apk_device_copy", "apk_ntt",
    "apk_prove", "apk_prove_device", "apk_verify", "apk_verify_ex", "apk_g2_decompress", "apk_g2_mul_generator", "apk_g1_mul_batch", "apk_g1_decompress", "apk_g1_to_lagrange", "apk_marshal_proof", "apk_marshal_public_inputs",
    "apk_fe_from_be", "apk_fe_to_be", "apk_hash_fr", "apk_host_fe_op", "apk_host_g1_op", "apk_g1_sum",
    "apk_device_alloc", "apk_device_free", "apk_device_upload", "apk_device_download",
    "apk_host_alloc", "apk_host_free", "apk_host_register", "apk_host_unregister",
    "apk_stats_enable", "apk_stats_read", "apk_paths_read",
    "apk_ctx_set_wire_hook", "apk_coset_ntt_device",
    "apk_comm_create", "apk_comm_destroy", "apk_comm_rank", "apk_comm_world", "apk_comm_barrier", "apk_comm_max_f64", "apk_comm_bind",
    "apk_comm_transport", "apk_msm_g1_sharded", "apk_comm_split_begin", "apk_comm_split_end", "apk_comm_serve", "apk_comm_set_compute",
    "apk_comm_commit", "apk_comm_wires", "apk_comm_rccl_ranks", "apk_comm_rccl_selftest",
    "apk_comm_commit_local", "apk_comm_spmd_begin", "apk_comm_spmd_end", "apk_comm_allgather_device", "apk_comm_subcoset_active",
    "apk_ctx_set_subcoset", "apk_comm_transport_reason", "apk_comm_link_probe", "apk_comm_phase_ms", "apk_ctx_msm_window",
]


class ApkError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__("libapk error %d: %s" % (code, msg))
        self.code = code


class CircuitDesc(C.Structure):
    _fields_ = [
        ("curve", C.c_int), ("device", C.c_int), ("n", C.c_uint64), ("nb_public", C.c_uint32),
        ("nb_commitments", C.c_uint32), ("srs_g1", C.c_void_p), ("srs_g1_lagrange", C.c_void_p),
        ("ql", C.c_void_p), ("qr", C.c_void_p), ("qm", C.c_void_p), ("qo", C.c_void_p), ("qk", C.c_void_p),
        ("perm", C.c_void_p), ("qcp", C.c_void_p * MAX_COMMITMENTS),
        ("commitment_constraint_index", C.c_uint32 * MAX_COMMITMENTS), ("msm_window", C.c_int), ("slots", C.c_int),
    ]


class Vk(C.Structure):
    _fields_ = [
        ("ql", C.c_uint8 * G1_MAX), ("qr", C.c_uint8 * G1_MAX), ("qm", C.c_uint8 * G1_MAX), ("qo", C.c_uint8 * G1_MAX),
        ("qk", C.c_uint8 * G1_MAX), ("s", (C.c_uint8 * G1_MAX) * 3), ("qcp", (C.c_uint8 * G1_MAX) * MAX_COMMITMENTS),
        ("size_inv", C.c_uint8 * 32), ("generator", C.c_uint8 * 32), ("coset_shift", C.c_uint8 * 32),
    ]


class Proof(C.Structure):
    _fields_ = [
        ("curve", C.c_uint32), ("nb_commitments", C.c_uint32),
        ("lro", (C.c_uint8 * G1_MAX) * 3), ("z", C.c_uint8 * G1_MAX), ("h", (C.c_uint8 * G1_MAX) * 3),
        ("bsb22", (C.c_uint8 * G1_MAX) * MAX_COMMITMENTS), ("batched_h", C.c_uint8 * G1_MAX),
        ("claimed_values", (C.c_uint8 * 32) * (6 + MAX_COMMITMENTS)), ("zshift_h", C.c_uint8 * G1_MAX),
        ("zshift_value", C.c_uint8 * 32),
        ("gamma", C.c_uint8 * 32), ("beta", C.c_uint8 * 32), ("alpha", C.c_uint8 * 32), ("zeta", C.c_uint8 * 32),
        ("gamma_kzg", C.c_uint8 * 32),
    ]


class VerifyingKey(C.Structure):
    _fields_ = [
        ("curve", C.c_int), ("n", C.c_uint64), ("nb_public", C.c_uint32), ("nb_commitments", C.c_uint32),
        ("ql", C.c_uint8 * G1_MAX), ("qr", C.c_uint8 * G1_MAX), ("qm", C.c_uint8 * G1_MAX), ("qo", C.c_uint8 * G1_MAX),
        ("qk", C.c_uint8 * G1_MAX), ("s", (C.c_uint8 * G1_MAX) * 3), ("qcp", (C.c_uint8 * G1_MAX) * MAX_COMMITMENTS),
        ("commitment_constraint_index", C.c_uint32 * MAX_COMMITMENTS), ("g1", C.c_uint8 * G1_MAX),
        ("g2", (C.c_uint8 * G2_MAX) * 2),
    ]


class VerifyTrace(C.Structure):
    _fields_ = [
        ("gamma", C.c_uint8 * 32), ("beta", C.c_uint8 * 32), ("alpha", C.c_uint8 * 32), ("zeta", C.c_uint8 * 32),
        ("pi", C.c_uint8 * 32), ("lin_at_zeta", C.c_uint8 * 32), ("gamma_kzg", C.c_uint8 * 32), ("folded_claim", C.c_uint8 * 32),
        ("lin_commitment", C.c_uint8 * G1_MAX), ("folded_digest", C.c_uint8 * G1_MAX),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("msm_accumulate_ms", C.c_double), ("msm_accumulate_launches", C.c_uint64), ("msm_pairs", C.c_uint64),
        ("msm_total_ms", C.c_double), ("msm_batches", C.c_uint64), ("ntt_ms", C.c_double), ("ntt_elements", C.c_uint64),
        ("prove_ms", C.c_double), ("proofs", C.c_uint64), ("round_ms", C.c_double * 4), ("host_lincomb_ms", C.c_double),
        ("msm_sort_ms", C.c_double), ("msm_tail_ms", C.c_double),
    ]


class PathCounts(C.Structure):
    """apk_path_counts: which forms of the load-dependent kernels ran (include/apk.h)."""
    _names = ["proofs", "msm_batches", "msm_sort_two_level", "msm_sort_two_level_by_load", "msm_sort_fused", "msm_lean_tail",
              "msm_rowcol_serial", "msm_combine_quad", "msm_small_units", "msm_one_launch", "msm_lagrange_wires", "ntt_sequences",
              "ntt_radix4", "ntt_radix4_by_load", "tail_fill_proofs", "host_lincomb_pooled", "msm_units_by_load", "host_inputs",
              "gang_proofs", "gang_msm_launches", "gang_ntt_launches", "gang_kernel_launches"]
    _fields_ = [(n, C.c_uint64) for n in _names] + [("reserved", C.c_uint64 * 2)]

    def as_dict(self) -> dict:
        return {n: int(getattr(self, n)) for n in self._names}


# int hook(void* user, int basis, uint32_t count, const void* const* d_scalars, const uint32_t* lens, void* out_points)
COMMIT_HOOK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.c_void_p)


# int hook(void* user, uint32_t count, const void* const* d_canonical, const uint32_t* lens, void* const* d_evals)
WIRE_HOOK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.POINTER(C.c_void_p))

# apk_compute: the GPU touch points of the communicator's schedules (test seam, include/apk.h)
CP_MSM = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_void_p)
CP_COSET = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p)
CP_ALLOC = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p))
CP_RELEASE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p)
CP_COPY = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int)


class Compute(C.Structure):
    _fields_ = [("user", C.c_void_p), ("msm_batch", CP_MSM), ("coset_ntt", CP_COSET), ("alloc", CP_ALLOC), ("release", CP_RELEASE),
                ("copy", CP_COPY), ("g1_bytes", C.c_size_t), ("n", C.c_uint64)]


def _load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libapk.so not found at %s - build it with `make -C algoplonk_amd/csrc` "
            "(there is no CPU fallback for the HIP path)" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp, u64, i32, sz = C.c_void_p, C.c_uint64, C.c_int, C.c_size_t
    lib.apk_last_error.restype = C.c_char_p
    lib.apk_abi_version.restype = i32
    lib.apk_device_count.argtypes = [C.POINTER(i32)]
    lib.apk_g1_bytes.argtypes = [i32]; lib.apk_g1_bytes.restype = sz
    lib.apk_fp_bytes.argtypes = [i32]; lib.apk_fp_bytes.restype = sz
    lib.apk_ctx_create.argtypes = [C.POINTER(CircuitDesc), C.POINTER(vp)]
    lib.apk_ctx_destroy.argtypes = [vp]; lib.apk_ctx_destroy.restype = None
    lib.apk_msm_ctx_create.argtypes = [i32, i32, vp, u64, i32, C.POINTER(vp)]
    lib.apk_ctx_get_vk.argtypes = [vp, C.POINTER(Vk)]
    lib.apk_msm_g1.argtypes = [vp, i32, vp, u64, vp]
    lib.apk_msm_g1_device.argtypes = [vp, i32, vp, u64, vp]
    lib.apk_msm_g1_batch_device.argtypes = [vp, i32, C.c_uint32, vp, vp, vp, vp]
    lib.apk_ctx_set_commit_hook.argtypes = [vp, COMMIT_HOOK, vp]
    lib.apk_device_copy.argtypes = [vp, vp, vp, sz]
    lib.apk_ntt.argtypes = [vp, i32, i32, i32, vp]
    lib.apk_prove.argtypes = [vp, vp, vp, vp, vp, vp, vp, C.POINTER(Proof)]
    lib.apk_prove_device.argtypes = [vp, vp, vp, vp, vp, vp, vp, C.POINTER(Proof)]
    lib.apk_verify.argtypes = [C.POINTER(VerifyingKey), C.POINTER(Proof), vp]
    lib.apk_verify_ex.argtypes = [C.POINTER(VerifyingKey), C.POINTER(Proof), vp, C.c_uint32, C.POINTER(VerifyTrace)]
    lib.apk_g2_decompress.argtypes = [i32, vp, vp]
    lib.apk_g2_mul_generator.argtypes = [i32, vp, vp]
    lib.apk_g1_mul_batch.argtypes = [i32, i32, vp, vp, u64, vp]
    lib.apk_g1_decompress.argtypes = [i32, i32, vp, u64, vp]
    lib.apk_g1_to_lagrange.argtypes = [i32, i32, vp, u64, vp]
    lib.apk_marshal_proof.argtypes = [C.POINTER(Proof), vp, sz, C.POINTER(sz)]
    lib.apk_marshal_public_inputs.argtypes = [i32, vp, C.c_uint32, vp, sz]
    lib.apk_fe_from_be.argtypes = [i32, i32, vp, vp]
    lib.apk_fe_to_be.argtypes = [i32, i32, vp, vp]
    lib.apk_hash_fr.argtypes = [i32, vp, vp]
    lib.apk_host_fe_op.argtypes = [i32, i32, i32, vp, vp, vp]
    lib.apk_host_g1_op.argtypes = [i32, i32, vp, vp, vp]
    lib.apk_g1_sum.argtypes = [i32, vp, u64, vp]
    lib.apk_device_alloc.argtypes = [vp, sz, C.POINTER(vp)]
    lib.apk_device_free.argtypes = [vp, vp]
    lib.apk_device_upload.argtypes = [vp, vp, vp, sz]
    lib.apk_device_download.argtypes = [vp, vp, vp, sz]
    lib.apk_host_alloc.argtypes = [i32, sz, C.POINTER(vp)]
    lib.apk_host_free.argtypes = [vp]
    lib.apk_host_register.argtypes = [vp, sz]
    lib.apk_host_unregister.argtypes = [vp]
    lib.apk_stats_enable.argtypes = [vp, i32]
    lib.apk_stats_read.argtypes = [vp, C.POINTER(Stats), i32]
    lib.apk_paths_read.argtypes = [vp, C.POINTER(PathCounts), i32]
    lib.apk_ctx_msm_window.argtypes = [vp]
    lib.apk_ctx_set_wire_hook.argtypes = [vp, WIRE_HOOK, vp]
    lib.apk_coset_ntt_device.argtypes = [vp, vp, u64, vp]
    lib.apk_comm_create.argtypes = [i32, i32, C.c_char_p, i32, C.POINTER(vp)]
    lib.apk_comm_destroy.argtypes = [vp]; lib.apk_comm_destroy.restype = None
    lib.apk_comm_rank.argtypes = [vp]
    lib.apk_comm_world.argtypes = [vp]
    lib.apk_comm_barrier.argtypes = [vp]
    lib.apk_comm_max_f64.argtypes = [vp, C.POINTER(C.c_double)]
    lib.apk_comm_bind.argtypes = [vp, vp]
    lib.apk_comm_transport.argtypes = [vp]; lib.apk_comm_transport.restype = C.c_char_p
    lib.apk_comm_rccl_ranks.argtypes = [vp]
    lib.apk_comm_transport_reason.argtypes = [vp]; lib.apk_comm_transport_reason.restype = C.c_char_p
    lib.apk_comm_link_probe.argtypes = [vp, sz, sz, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.apk_comm_phase_ms.argtypes = [vp, C.POINTER(C.c_double), i32]
    lib.apk_comm_rccl_selftest.argtypes = [i32, C.POINTER(i32)]
    lib.apk_msm_g1_sharded.argtypes = [vp, vp, u64, vp]
    lib.apk_comm_split_begin.argtypes = [vp]
    lib.apk_comm_split_end.argtypes = [vp]
    lib.apk_comm_serve.argtypes = [vp, C.POINTER(u64)]
    lib.apk_comm_set_compute.argtypes = [vp, C.POINTER(Compute)]
    lib.apk_comm_commit.argtypes = [vp, i32, C.c_uint32, vp, vp, vp]
    lib.apk_comm_wires.argtypes = [vp, C.c_uint32, vp, vp, vp]
    lib.apk_comm_commit_local.argtypes = [vp, i32, C.c_uint32, vp, vp, vp]
    lib.apk_comm_spmd_begin.argtypes = [vp]
    lib.apk_comm_spmd_end.argtypes = [vp]
    lib.apk_comm_allgather_device.argtypes = [vp, vp, C.c_size_t]
    lib.apk_comm_subcoset_active.argtypes = [vp]
    lib.apk_ctx_set_subcoset.argtypes = [vp, i32, i32, vp, vp]
    return lib


lib = _load()


def check(code: int) -> None:
    if code != APK_OK:
        raise ApkError(code, (lib.apk_last_error() or b"").decode())


def device_count() -> int:
    n = C.c_int(0)
    check(lib.apk_device_count(C.byref(n)))
    return n.value
