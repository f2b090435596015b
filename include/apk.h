/* libapk - C-ABI of the MI355X-native PLONK prover path for AlgoPlonk.
 *
 * Drop-in boundary.  The reference has no FFI of its own: the hot path is the Go call
 *     proof, err := plonk.Prove(cc.Ccs, cc.Pk, witness)         /root/reference/algoplonk.go:89
 * (second call site /root/reference/testutils/testutils.go:47) and the one-time
 *     plonk.Setup(ccs, srs, lagrangeSrs)                         /root/reference/setup/setup.go:107,149
 * A cgo shim replaces those calls with the entry points below.  Every buffer uses
 * gnark's in-memory layout so the shim passes unsafe.Pointer(&slice[0]) with zero copies:
 *   - Fr / Fp element : little-endian limbs, Montgomery form (gnark-crypto fr.Element / fp.Element);
 *                       32 bytes for Fr (both curves), 32 (BN254) / 48 (BLS12-381) bytes for Fp
 *   - G1 affine       : X || Y, each an Fp element as above; (0,0) is the point at infinity
 * No torch types, no C++ types: plain pointers and sizes.  All functions return APK_OK (0) or an error
 * code; apk_last_error() gives the message for the calling thread.  Nothing panics or aborts across the
 * boundary (the reference wraps errors with fmt.Errorf, algoplonk.go:90-92).
 *
 * There is NO CPU fallback: every compute entry point fails with APK_ERR_HIP when no gfx950 device is
 * usable.  The oracle under oracle/ is test infrastructure and is never linked into this library.
 *
 * Runtime environment: a context runs up to 16 HIP streams at a time - one proof each, or (circuits up to 2^16, more callers
 * than streams) a gang of two to four proofs that share a stream and its launches.  ROCm maps streams onto
 * GPU_MAX_HW_QUEUES hardware queues (default 4), read ONCE when the HIP runtime initialises; with 4 the streams serialise
 * (-20 % proofs/s measured).  The library therefore sets GPU_MAX_HW_QUEUES=24 when it is loaded, unless the variable is
 * already set.  A host process that initialises HIP BEFORE loading libapk (another GPU library, torch ...) must export
 * GPU_MAX_HW_QUEUES=24 itself before its first HIP call - the cgo shim of INTEGRATION.md loads libapk at program start, so
 * a plain AlgoPlonk process needs nothing.
 */
#ifndef APK_H
#define APK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define APK_ABI_VERSION 5

/* curve ids: the two curves the AVM supports (algoplonk.go:39-41) */
#define APK_BN254 0
#define APK_BLS12_381 1

#define APK_OK 0
#define APK_ERR_ARG 1      /* bad argument (unknown curve, n not a power of two, null pointer ...) */
#define APK_ERR_HIP 2      /* HIP runtime failure, including "no device" */
#define APK_ERR_STATE 3    /* call not valid for this context (e.g. Lagrange MSM without a Lagrange SRS) */
#define APK_ERR_WITNESS 4  /* the witness does not satisfy the circuit (quotient not a polynomial) */
#define APK_ERR_VERIFY 5   /* apk_verify: the proof does not verify against the key and public inputs */

#define APK_FR_BYTES 32
#define APK_G1_MAX_BYTES 96       /* BLS12-381 affine; BN254 uses the first 64 bytes of a slot */
#define APK_G2_MAX_BYTES 192      /* G2 affine, gnark in-memory: X.A0 || X.A1 || Y.A0 || Y.A1; BN254 uses the first 128 bytes */
#define APK_MAX_COMMITMENTS 2     /* BSB22 commitments per circuit (reference documents 0/1/2: README.md:27-30) */
#define APK_NB_BLINDING 9         /* bl0,bl1, br0,br1, bo0,bo1, bz0,bz1,bz2 */

typedef struct apk_ctx apk_ctx;

const char* apk_last_error(void);
int apk_abi_version(void);
/* number of visible HIP devices (0 and APK_OK when the runtime loads but no GPU is present) */
int apk_device_count(int* count);
/* bytes of one G1 affine point / one Fp element for a curve id; 0 for an unknown id */
size_t apk_g1_bytes(int curve);
size_t apk_fp_bytes(int curve);

/* ---- circuit context: replaces plonk.Setup's output + the per-proof trace rebuild ----------------------
 * One context per (device, curve, circuit).  It owns, resident in HBM: the KZG SRS with its windowed point
 * tables, the trace polynomials (Lagrange, canonical and 4n-coset forms), the permutation polynomials, the NTT
 * twiddles and `slots` independent proving workspaces.  Inputs are copied; the caller keeps ownership
 * (SURVEY.md §8b "nothing is retained by the callee").
 */
typedef struct {
    int curve;                 /* APK_BN254 / APK_BLS12_381 */
    int device;                /* HIP device ordinal */
    uint64_t n;                /* domain size: NextPowerOfTwo(nbConstraints + nbPublic), >= 8 (setup/setup.go:113-114) */
    uint32_t nb_public;        /* VK NbPublicVariables */
    uint32_t nb_commitments;   /* BSB22 commitments (<= APK_MAX_COMMITMENTS) */
    const void* srs_g1;        /* n+3 G1 affine: canonical SRS  = gnark kzg.ProvingKey.G1 (setup/setup.go:113-114) */
    const void* srs_g1_lagrange; /* n G1 affine: Lagrange SRS = gnark's ProvingKey.KzgLagrange (setup/setup.go:124,138), or NULL: the
                                  * context then derives it from srs_g1 on the device (kzg.ToLagrangeG1) unless APK_WIRES_LAGRANGE=0
                                  * and nb_commitments == 0.  It serves the BSB22 commitments and - round 5 - [L][R][O], which gnark
                                  * commits over this basis: the scalars are then the witness values themselves (mostly 0 / 1 / small
                                  * in real circuits: few non-zero digits).  APK_WIRES_LAGRANGE (read at apk_ctx_create): -1 = the
                                  * context decides per proof from the wires' non-zero digits (default), 0 = always the canonical
                                  * SRS (rounds 1-4), 1 = always the Lagrange SRS.  Same group elements, same proof bytes. */
    const void* ql;            /* trace columns in Lagrange form, n Fr each  = gnark plonk.Trace{Ql,Qr,Qm,Qo,Qk} */
    const void* qr;
    const void* qm;
    const void* qo;
    const void* qk;            /* public rows zero */
    const int64_t* perm;       /* 3n entries = gnark Trace.S */
    const void* qcp[APK_MAX_COMMITMENTS];               /* n Fr each */
    uint32_t commitment_constraint_index[APK_MAX_COMMITMENTS]; /* VK CommitmentConstraintIndexes */
    int msm_window;            /* signed-digit window bits (7..20; 18..20 = 2^17..2^19 buckets, two-level sort only: needs
                                * bases x windows < 2^(31 - partition bits)); 0 = choose from n and slots */
    int slots;                 /* concurrent proofs in flight on this context; 0 = 1; capped at 16 (more callers wait their turn) */
} apk_circuit_desc;

int apk_ctx_create(const apk_circuit_desc* desc, apk_ctx** out);
/* the window width the context chose (msm_window = 0) or was given; 0 for a null context */
int apk_ctx_msm_window(apk_ctx* ctx);
void apk_ctx_destroy(apk_ctx* ctx);
/* MSM-only context over an arbitrary base set (`count` G1 affine, host memory): windowed tables + one MSM
 * workspace, no circuit.  Used to shard ONE large MSM by index range across GPUs (BASELINE.json configs[3]):
 * each rank builds a context over its slice of the SRS, runs apk_msm_g1 on its slice of the scalars, and the
 * partial sums (one point per rank) are all-gathered and added (algoplonk_amd/parallel.py).  apk_msm_g1* with
 * basis 0 and the device-memory helpers work on it; apk_prove / apk_ntt / apk_ctx_get_vk return APK_ERR_STATE.
 * The bases must lie in the prime-order subgroup (every KZG SRS does): the tables hold R^-1 * P_i so that Montgomery-form scalars
 * are used as they arrive, and (a R)(R^-1 P) = a P holds only for points of order r.  BN254's G1 has cofactor 1; an on-curve
 * BLS12-381 point outside the subgroup gives a different sum than gnark's MultiExp (not checked: a subgroup test per base would
 * double the table build). */
int apk_msm_ctx_create(int curve, int device, const void* bases, uint64_t count, int msm_window, apk_ctx** out);

/* Verifying-key commitments produced during context creation (the 8+k MSMs of plonk.Setup).
 * Each slot is APK_G1_MAX_BYTES wide, gnark in-memory affine form. Order: Ql,Qr,Qm,Qo,Qk,S1,S2,S3,Qcp_0.. */
typedef struct {
    uint8_t ql[APK_G1_MAX_BYTES], qr[APK_G1_MAX_BYTES], qm[APK_G1_MAX_BYTES], qo[APK_G1_MAX_BYTES], qk[APK_G1_MAX_BYTES];
    uint8_t s[3][APK_G1_MAX_BYTES];
    uint8_t qcp[APK_MAX_COMMITMENTS][APK_G1_MAX_BYTES];
    uint8_t size_inv[APK_FR_BYTES], generator[APK_FR_BYTES], coset_shift[APK_FR_BYTES]; /* Fr, Montgomery */
} apk_vk;
int apk_ctx_get_vk(apk_ctx* ctx, apk_vk* out);

/* ---- primitives (row a4 / a6 of SURVEY.md §8a) --------------------------------------------------------- */
/* kzg.Commit: sum scalars[i] * SRS[i].  basis 0 = canonical SRS (len <= n+3), 1 = Lagrange SRS (len <= n; the context's table
 * continues with the three blinding points [tau^(n+k)]G1 - [tau^k]G1 at indices n..n+2, so len <= n+3 is accepted).
 * scalars: host memory, `len` Fr in Montgomery form.  out: one G1 affine (apk_g1_bytes). */
int apk_msm_g1(apk_ctx* ctx, int basis, const void* scalars, uint64_t len, void* out);
/* same, scalars already resident in device memory (what bench.py times: inputs in HBM) */
int apk_msm_g1_device(apk_ctx* ctx, int basis, const void* d_scalars, uint64_t len, void* out);
/* `count` (<= 4) partial commitments in one launch sequence: out[b] = sum_{i < lens[b]} d_scalars[b][i] * SRS[offsets[b] + i].
 * The building block of an MSM split by index range over GPUs that each hold the whole SRS (algoplonk_amd/parallel.py
 * SplitCommitter): a rank commits the slices it was dealt, the partial sums are all-gathered and added (apk_g1_sum). */
int apk_msm_g1_batch_device(apk_ctx* ctx, int basis, uint32_t count, const void* const* d_scalars, const uint64_t* offsets,
                            const uint64_t* lens, void* out);
/* fft.Domain.FFT / FFTInverse on the context's size-n (which=0) or size-4n (which=1) domain, natural order in
 * and out; coset != 0 evaluates on / interpolates from the coset CosetShift * <omega>.  data: host, in place. */
int apk_ntt(apk_ctx* ctx, int which, int inverse, int coset, void* data);

/* ---- the prover: replaces plonk.Prove (algoplonk.go:89) ------------------------------------------------ */
typedef struct {
    uint32_t curve;
    uint32_t nb_commitments;
    uint8_t lro[3][APK_G1_MAX_BYTES];                      /* Proof.LRO */
    uint8_t z[APK_G1_MAX_BYTES];                           /* Proof.Z */
    uint8_t h[3][APK_G1_MAX_BYTES];                        /* Proof.H */
    uint8_t bsb22[APK_MAX_COMMITMENTS][APK_G1_MAX_BYTES];  /* Proof.Bsb22Commitments */
    uint8_t batched_h[APK_G1_MAX_BYTES];                   /* Proof.BatchedProof.H */
    uint8_t claimed_values[6 + APK_MAX_COMMITMENTS][APK_FR_BYTES]; /* Proof.BatchedProof.ClaimedValues: lin,l,r,o,s1,s2,qcp.. */
    uint8_t zshift_h[APK_G1_MAX_BYTES];                    /* Proof.ZShiftedOpening.H */
    uint8_t zshift_value[APK_FR_BYTES];                    /* Proof.ZShiftedOpening.ClaimedValue */
    /* diagnostics (not part of gnark's Proof): the Fiat-Shamir challenges, Fr Montgomery */
    uint8_t gamma[APK_FR_BYTES], beta[APK_FR_BYTES], alpha[APK_FR_BYTES], zeta[APK_FR_BYTES], gamma_kzg[APK_FR_BYTES];
} apk_proof;

/* Inputs (host memory, gnark layout): the solved wire columns L,R,O in Lagrange form (n Fr each) - what gnark's
 * solver leaves in `SparseR1CSSolution{L,R,O}`; the public inputs (nb_public Fr) = fullWitness[:nbPublic];
 * the 9 blinding scalars (APK_NB_BLINDING Fr) that gnark draws from crypto/rand - explicit here so proofs are
 * reproducible; pi2 = BSB22 committed columns (Lagrange, n Fr each, hiding entries placed) or NULL.
 * Blocks the calling thread; safe to call from several threads (each takes a free slot).
 * This is the call the cgo shim makes (algoplonk.go:89 hands plonk.Prove Go slices).  L, R, O (and pi2) travel
 * to the device on a copy stream of their own, started BEFORE the caller queues for a proving slot: with more callers than
 * slots the upload of the next proof runs beside the rounds of the proofs in flight.  The transfer is true asynchronous DMA
 * when the buffers are page-locked - allocate them with apk_host_alloc or register existing ones once with
 * apk_host_register (the shim does, per buffer pool); pageable buffers work too, staged by the HIP runtime on the calling
 * thread.  The buffers must stay unchanged until the call returns. */
int apk_prove(apk_ctx* ctx, const void* L, const void* R, const void* O, const void* public_inputs,
              const void* blinding, const void* const* pi2, apk_proof* out);

/* Variant with L,R,O already resident in device memory (n Fr each). */
int apk_prove_device(apk_ctx* ctx, const void* d_L, const void* d_R, const void* d_O, const void* public_inputs,
                     const void* blinding, const void* const* d_pi2, apk_proof* out);

/* Intra-proof multi-GPU.  With a hook installed, apk_prove* does not run its KZG commitments on
 * the context's own GPU: at each Fiat-Shamir sync point it calls the hook with the batch's scalar vectors (device memory of
 * this context, `lens[b]` Fr each, commitment b = sum d_scalars[b][i] * SRS_basis[i]) and expects `count` G1 affine points
 * (apk_g1_bytes each) in out_points.  The hook deals the work to the other GPUs of the node (algoplonk_amd/parallel.py:
 * scatter of scalar slices over RCCL, apk_msm_g1_batch_device on every rank, all-gather of the partial sums) while this GPU
 * runs the launches the prover queued behind the batch.  Return APK_OK or an error code.  NULL removes the hook. */
typedef int (*apk_commit_hook)(void* user, int basis, uint32_t count, const void* const* d_scalars, const uint32_t* lens, void* out_points);
int apk_ctx_set_commit_hook(apk_ctx* ctx, apk_commit_hook hook, void* user);
/* Same idea for the per-wire transforms of round 1 ("per-wire NTTs batch across the GPUs", BASELINE.json north_star): with a
 * wire hook installed the prover does not run the 4n-coset evaluations of l, r, o itself; it hands the hook the `count` blinded
 * canonical polynomials (device memory of this context, lens[i] Fr) and expects d_evals[i] (4n Fr each, device memory of this
 * context) filled when the hook returns.  apk_coset_ntt_device is what a rank runs on the polynomial it was dealt. */
typedef int (*apk_wire_hook)(void* user, uint32_t count, const void* const* d_canonical, const uint32_t* lens, void* const* d_evals);
int apk_ctx_set_wire_hook(apk_ctx* ctx, apk_wire_hook hook, void* user);
int apk_coset_ntt_device(apk_ctx* ctx, const void* d_canonical, uint64_t len, void* d_evals);
/* Sub-coset split of round 3, for the replicated prover: rank k of `world`
 * (2, 4 or 8) evaluates the wire and permutation polynomials only on the points i = k (mod world) of the 4n coset (a coset of
 * 4n / world points: one transform of that size per polynomial), runs the quotient kernel there and inverse-transforms locally;
 * ONE all-gather of 4n / world field elements per rank - the hook: `d_all` holds world x bytes_per_rank bytes of THIS context's
 * device memory, rank r's part at r * bytes_per_rank, this rank's part filled in; the hook returns with every part filled in -
 * and the last log2(world) butterfly stages on every rank give the quotient's 4n coefficients, bit for bit those of the whole-
 * coset path.  world = 1 or hook = NULL switches it off.  Needs a context whose Qk is completed inside the quotient kernel (the
 * default: at most 6 public inputs + commitments); APK_ERR_STATE otherwise. */
typedef int (*apk_gather_hook)(void* user, void* d_all, size_t bytes_per_rank);
int apk_ctx_set_subcoset(apk_ctx* ctx, int k, int world, apk_gather_hook hook, void* user);

/* ---- multi-GPU behind the boundary: one process per GPU, libapk's own communicator ---------------------
 * The reference's host is Go (algoplonk.go:89): a cgo caller cannot use a Python process group, so the exchange steps of the
 * path live here.  An apk_comm is one rank of `world` cooperating processes.
 *   control plane : a TCP star through rank 0 (rendezvous, headers, the 64/96-byte partial sums, barriers) - apk_comm_create;
 *   data plane    : RCCL over xGMI on libapk's OWN HIP runtime and stream (ncclSend/ncclRecv groups for the scatter of scalar
 *                   slices and the peer copies of polynomials; librccl is dlopen'ed by apk_comm_bind when world > 1, each rank
 *                   owns a different GPU and APK_COMM_RCCL is not 0); else HIP IPC (the sender exports its staging buffer, the
 *                   receiver maps it and pulls with one device-to-device copy: peer reads over xGMI between GPUs, on-device
 *                   copies when ranks share a GPU; APK_COMM_IPC=0 switches it off); else the same bytes are staged through the
 *                   host and the TCP star - what the CPU tier tests drive (two processes, no GPU).
 * Schedules (csrc/comm.cpp):
 *   apk_msm_g1_sharded : BASELINE configs[3] - ONE MSM split by index range: every rank commits its slice on the MSM-only
 *                        context bound to the communicator, ONE all-gather of a 64/96-byte point per rank, local additions;
 *                        the result is returned on every rank.  (A numeric all-reduce cannot add curve points.)
 *   split proof        : ONE proof, several GPUs.  The leader (rank 0) calls apk_comm_split_begin, proves as usual
 *                        (apk_prove*), then apk_comm_split_end; every other rank sits in apk_comm_serve.  Each commitment batch
 *                        is dealt by index range of its flattened (scalar, point) pairs: header broadcast, one scatter of the
 *                        scalar slices, apk_msm_g1_batch_device on every rank, all-gather of the partial sums; with
 *                        APK_SPLIT_WIRES=1 the 4n-coset evaluation of wire i is dealt to rank i mod world (peer copies of the
 *                        canonical polynomial out and of the evaluations back).
 * Every rank holds the circuit context (apk_ctx_create with the same inputs), so any rank can commit any index range.
 * Errors on one rank are reported on all ranks of the step (status words travel with the partial sums); a peer that stops
 * answering INSIDE a step fails the call after APK_COMM_TIMEOUT_S seconds (default 300); between steps a worker in apk_comm_serve
 * waits for the leader's next header without a timeout (the leader may be idle for any length of time).  When APK_COMM_TOKEN is
 * set, its hash travels in every rank's hello and rank 0 refuses connections that do not carry it.
 * With RCCL as the data plane the partial sums of a sharded MSM / a dealt commitment batch are exchanged with ncclAllGather
 * (north_star: "RCCL-over-xGMI ... for the final bucket-sum of a single MSM"); the ranks add the gathered points themselves. */
typedef struct apk_comm apk_comm;
int apk_comm_create(int rank, int world, const char* addr, int port, apk_comm** out);   /* rank 0 listens on addr:port */
void apk_comm_destroy(apk_comm* comm);
int apk_comm_rank(const apk_comm* comm);
int apk_comm_world(const apk_comm* comm);
int apk_comm_barrier(apk_comm* comm);
int apk_comm_max_f64(apk_comm* comm, double* value);       /* in place: maximum over the ranks (bench.py's timing protocol) */
/* This rank's context (a circuit context for the split proof, an MSM-only context for the sharded MSM) and the data plane.
 * Collective: every rank calls it.  The communicator allocates its staging buffers through the bound context, so the context
 * must outlive the binding: apk_comm_bind(comm, NULL) (not collective) or apk_comm_destroy BEFORE apk_ctx_destroy. */
int apk_comm_bind(apk_comm* comm, apk_ctx* ctx);
const char* apk_comm_transport(const apk_comm* comm);      /* "rccl", "ipc" or "tcp" (after apk_comm_bind) */
/* Size of the RCCL communicator behind the data plane (ncclCommCount): `world` when the transport is "rccl", 0 otherwise. */
int apk_comm_rccl_ranks(const apk_comm* comm);
/* Why the data plane is NOT RCCL after apk_comm_bind ("" when it is): "single rank", "APK_COMM_RCCL=0", "librccl.so could not be
 * loaded", "ranks q and r share device d", "RCCL bring-up failed ..." - bench.py prints it as "fallback:<why>" instead of silently
 * timing the TCP star. */
const char* apk_comm_transport_reason(const apk_comm* comm);
/* One timed pass over the data plane as it came up (collective, after apk_comm_bind on a device context): a ring of grouped
 * ncclSend / ncclRecv of ring_bytes (RCCL plane only; 0 elsewhere) and an in-place all-gather of gather_bytes per rank on the active
 * plane.  GB/s = bytes this rank received / wall time of the second of two passes.  This is how the first run on a multi-GPU node
 * reports its per-link rate (DESIGN.md section 6 prices its projections at 48 GB/s per xGMI link, unmeasured). */
int apk_comm_link_probe(apk_comm* comm, size_t ring_bytes, size_t gather_bytes, double* ring_gbps, double* gather_gbps);
/* Where a schedule's time went on this rank since the last reset, host wall clock: out[0] = this rank's commitment MSMs (ms),
 * out[1] = the partial sums' exchange, out[2] = the sub-coset all-gather, out[3] = commitment rounds, out[4] = gathers. */
int apk_comm_phase_ms(apk_comm* comm, double* out5, int reset);
/* A world-1 RCCL communicator on `device` driven through every RCCL call of the data plane (unique id, ncclCommInitRank, a grouped
 * ncclSend / ncclRecv to itself on a non-blocking stream, ncclAllGather, ncclCommCount, ncclCommDestroy), results checked byte for
 * byte.  One-GPU boxes cannot run two RCCL ranks (RCCL refuses two ranks per device): this is how that branch's init, stream
 * use and teardown execute on hardware there.  *ranks receives ncclCommCount (1). */
int apk_comm_rccl_selftest(int device, int* ranks);
int apk_msm_g1_sharded(apk_comm* comm, const void* d_scalars, uint64_t len, void* out);
int apk_comm_split_begin(apk_comm* comm);
int apk_comm_split_end(apk_comm* comm);
int apk_comm_serve(apk_comm* comm, uint64_t* steps_served);
/* Test seam: the GPU touch points of the schedules as a table, so that the CPU tier can run the SAME C code (transport, dealing,
 * error propagation) in two processes without a GPU - "device" pointers are then host pointers and the table's msm is the
 * oracle's.  NULL entries keep the built-in implementation (the bound context).  kind: 0 device-to-device, 1 host-to-device,
 * 2 device-to-host. */
typedef struct {
    void* user;
    int (*msm_batch)(void* user, int basis, uint32_t count, const void* const* d_scalars, const uint64_t* offsets,
                     const uint64_t* lens, void* out_points);
    int (*coset_ntt)(void* user, const void* d_in, uint64_t len, void* d_out);
    int (*alloc)(void* user, size_t bytes, void** d_ptr);
    int (*release)(void* user, void* d_ptr);
    int (*copy)(void* user, void* dst, const void* src, size_t bytes, int kind);
    size_t g1_bytes;          /* 64 / 96 */
    uint64_t n;               /* domain size (coset evaluations are 4n Fr) */
} apk_compute;
int apk_comm_set_compute(apk_comm* comm, const apk_compute* table);
/* the schedules' entry points as the hooks call them (exposed for the tests of the seam) */
int apk_comm_commit(apk_comm* comm, int basis, uint32_t count, const void* const* d_scalars, const uint32_t* lens, void* out_points);
int apk_comm_wires(apk_comm* comm, uint32_t count, const void* const* d_canonical, const uint32_t* lens, void* const* d_evals);
/* Replicated prover ("SPMD", round 4): every rank holds the circuit context AND the witness and makes the same apk_prove* call
 * between apk_comm_spmd_begin and apk_comm_spmd_end (both called on every rank).  Only the commitments are shared out: rank r
 * commits its index range of each batch from its OWN copy of the polynomials, one all-gather of the partial sums, every rank adds
 * them - nothing is scattered (the leader / worker split above moves 604 MB per BLS12-381 2^21 proof out of rank 0) and the
 * Fiat-Shamir transcripts of the ranks stay identical, so every rank returns the same proof.  One proof at a time per
 * communicator.  apk_comm_commit_local is the step the hook runs (exposed for the tests of the seam). */
/* The sub-coset split of round 3 is ON by default in this mode for worlds of 2, 4 and 8 (APK_SPMD_SUBCOSET=0 switches it off) on
 * the strength of ELEMENT COUNTS only (24 n transformed elements per proof against 4 n + 20 n / G) and a projected 48 GB/s per xGMI
 * link: no box of the build had two GPUs.  bench.py --mode prove-spmd prints the measured link rate (apk_comm_link_probe) and the
 * per-phase times (apk_comm_phase_ms) so that the first multi-GPU run can confirm or overturn that default. */
int apk_comm_spmd_begin(apk_comm* comm);
int apk_comm_spmd_end(apk_comm* comm);
int apk_comm_commit_local(apk_comm* comm, int basis, uint32_t count, const void* const* d_scalars, const uint32_t* lens, void* out_points);
/* In-place all-gather of device memory (the sub-coset split's one exchange): d_all holds world x bytes_per_rank bytes, rank r's
 * part at r * bytes_per_rank.  ncclAllGather on the RCCL plane, pulls from the peers' exported buffers on the IPC plane, staged
 * through the host and the TCP star otherwise.  apk_comm_spmd_begin installs it as the bound context's gather hook when the world
 * is 2, 4 or 8 and APK_SPMD_SUBCOSET is not 0. */
int apk_comm_allgather_device(apk_comm* comm, void* d_all, size_t bytes_per_rank);
int apk_comm_subcoset_active(const apk_comm* comm);        /* 1 between spmd_begin and spmd_end when round 3 runs on sub-cosets */

/* ---- the verifier: the host-side mirror of plonk.Verify(proof, vk, publicWitness) (algoplonk.go:93) --------------------
 * (*CompiledCircuit).Verify runs the prover AND gnark's verifier before it hands out a VerifiedProof (algoplonk.go:79-98).
 * A Go integration keeps calling gnark's plonk.Verify on the returned *Proof; hosts without gnark (algoplonk_amd's Python
 * mirror of the API) call apk_verify.  Host only, no GPU: transcript, ~25 G1 scalar multiplications, one two-pair pairing
 * check.  The statement checked is the one the reference's generated AVM verifiers check
 * (verifier/templateLogicSigBN254.go:110-356).  Returns APK_OK, APK_ERR_VERIFY (proof rejected) or APK_ERR_ARG (bad key). */
typedef struct {
    int curve;
    uint64_t n;                /* VK Size */
    uint32_t nb_public;        /* VK NbPublicVariables */
    uint32_t nb_commitments;
    uint8_t ql[APK_G1_MAX_BYTES], qr[APK_G1_MAX_BYTES], qm[APK_G1_MAX_BYTES], qo[APK_G1_MAX_BYTES], qk[APK_G1_MAX_BYTES];
    uint8_t s[3][APK_G1_MAX_BYTES];
    uint8_t qcp[APK_MAX_COMMITMENTS][APK_G1_MAX_BYTES];
    uint32_t commitment_constraint_index[APK_MAX_COMMITMENTS];
    uint8_t g1[APK_G1_MAX_BYTES];        /* VK Kzg.G1 = SRS G1[0] */
    uint8_t g2[2][APK_G2_MAX_BYTES];     /* VK Kzg.G2 = ([1]G2, [tau]G2), gnark in-memory G2Affine */
} apk_verifying_key;
int apk_verify(const apk_verifying_key* vk, const apk_proof* proof, const void* public_inputs);   /* public_inputs: vk->nb_public Fr */
/* Same check with (a) the length of the public witness stated by the caller - gnark's plonk.Verify rejects
 * len(publicWitness) != vk.NbPublicVariables, and so does this call (APK_ERR_VERIFY) instead of reading vk->nb_public values
 * from a shorter buffer - and (b) optionally the intermediate values of the verification under the names the reference's
 * AVM template gives them (verifier/templateLogicSigBN254.go:137-140 gamma/beta/alpha/zeta, :181-193 PI, :218
 * linearized_poly_at_z, :256-278 lin_poly_com, :281-287 the folding challenge, :289-320 digest/claims before the two openings are
 * batched).  Scalars canonical big-endian; points X || Y big-endian as the AVM's ec ops return them (all zero = infinity).
 * tests/test_template_pin.py compares them with the values the executed template produced (tests/golden/template_verdicts.json).
 * Fields behind the point of rejection stay zero. */
typedef struct {
    uint8_t gamma[APK_FR_BYTES], beta[APK_FR_BYTES], alpha[APK_FR_BYTES], zeta[APK_FR_BYTES];
    uint8_t pi[APK_FR_BYTES], lin_at_zeta[APK_FR_BYTES], gamma_kzg[APK_FR_BYTES], folded_claim[APK_FR_BYTES];
    uint8_t lin_commitment[APK_G1_MAX_BYTES], folded_digest[APK_G1_MAX_BYTES];
} apk_verify_trace;
int apk_verify_ex(const apk_verifying_key* vk, const apk_proof* proof, const void* public_inputs, uint32_t nb_public_inputs,
                  apk_verify_trace* trace /* may be NULL */);
/* G2 points for apk_verifying_key.g2 (host only).  apk_g2_decompress: one compressed G2 exactly as it sits in the
 * reference's vk.bin files (64 | 96 bytes, X.A1 || X.A0 big-endian with gnark's flag bits) -> in-memory form.
 * apk_g2_mul_generator: [scalar]G2 - the G2 side of a TestOnly SRS whose tau is known (unsafekzg, setup/setup.go:102-108);
 * scalar = one Fr in Montgomery form. */
int apk_g2_decompress(int curve, const uint8_t* compressed, void* out);
int apk_g2_mul_generator(int curve, const void* scalar_fr, void* out);

/* ---- test-only SRS generation: the device half of gnark's test/unsafekzg.NewSRS (setup/setup.go:102-108) ---
 * out[i] = scalars[i] * base.  Host buffers; scalars Fr Montgomery; base/out G1 affine.  The caller supplies
 * tau^i (canonical SRS) or L_i(tau) (Lagrange SRS) as the scalars. */
int apk_g1_mul_batch(int curve, int device, const void* base, const void* scalars, uint64_t count, void* out);

/* ---- trusted-setup loading on the GPU: what setup.Run's trusted branch does before plonk.Setup ----
 * apk_g1_decompress : kzg SRS ReadFrom (setup/setup.go:173-174,189-190).  `compressed` = count x (32 | 48) bytes exactly as
 *                     they sit in pk.bin after the 4-byte count (big-endian X with gnark's flag bits);
 *                     out = count G1 affine in gnark's in-memory form.  APK_ERR_ARG if any encoding is not a curve point.
 * apk_g1_to_lagrange: kzg.ToLagrangeG1 (setup/setup.go:124,138): out[i] = [L_i(tau)]G1 from points[j] = [tau^j]G1, n a
 *                     power of two, without knowing tau (inverse FFT in the exponent).  Host buffers. */
int apk_g1_decompress(int curve, int device, const uint8_t* compressed, uint64_t count, void* out);
int apk_g1_to_lagrange(int curve, int device, const void* points, uint64_t n, void* out);

/* ---- wire formats (host only, no GPU needed): replace MarshalProof / MarshalPublicInputs ----------------- */
/* helper.go:13-24,27-88: 768 + 96k bytes (BN254) / 1056 + 128k bytes (BLS12-381).  Returns the length in *len. */
int apk_marshal_proof(const apk_proof* proof, uint8_t* out, size_t cap, size_t* len);
/* helper.go:91-110: nb_public * 32 big-endian canonical bytes. */
int apk_marshal_public_inputs(int curve, const void* public_inputs, uint32_t nb_public, uint8_t* out, size_t cap);
/* Montgomery <-> canonical big-endian conversions for Fr (field=0) and Fp (field=1) elements; host only. */
int apk_fe_from_be(int curve, int field, const uint8_t* be, void* out);
int apk_fe_to_be(int curve, int field, const void* in, uint8_t* be);
/* hash_to_field with DST "BSB22-Plonk" over a marshalled G1 point (templateLogicSigBN254.go:386-397); host only. */
int apk_hash_fr(int curve, const void* g1_affine, void* out_fr);

/* ---- diagnostics: run the library's own field / curve templates on the HOST (no GPU).  The kernels are built
 * from the same templates, so the CPU-only test tier can pin the formulas against the oracle.
 * field ops: 0 add, 1 sub, 2 mul (Montgomery), 3 inverse, 4 neg; 10-13 = mul / add / sub / neg on unsaturated limbs; 14 = ten
 * lazy butterfly stages (u, v) <- (u + b v, u - b v) from (a, b) as the NTT tile runs them, returns u.  field: 0 = Fr, 1 = Fp.
 * g1 ops: 0 mixed add p+q, 1 full XYZZ add p+q, 2 double p, 3 scalar mul q*p (q = Fr Montgomery); 10/11 = 0/1 on the
 * MSM's unsaturated limbs; 12/13 = a fixed 17-step signed chain ending at p+3q through the accumulate loop's lazy
 * mixed addition / the plain one; 14 = lazy full additions and doublings ending at 4p+6q. */
int apk_host_fe_op(int curve, int field, int op, const void* a, const void* b, void* out);
int apk_host_g1_op(int curve, int op, const void* p, const void* q, void* out);
/* out = sum of `count` G1 affine points (host; 0 points give infinity): the local half of a sharded MSM's one exchange
 * step - the all-gathered per-rank partial sums are added with this call (algoplonk_amd/parallel.py). */
int apk_g1_sum(int curve, const void* points, uint64_t count, void* out);

/* ---- page-locked host memory for the inputs of apk_prove -------------------------------------------------------------------
 * apk_host_alloc: `bytes` of page-locked host memory, visible to every device (hipHostMalloc, portable); apk_host_free.
 * apk_host_register / apk_host_unregister: page-lock an EXISTING buffer for the lifetime of the registration (hipHostRegister) -
 * what a cgo host does once per witness-buffer pool: Go's heap does not move objects, and the pointer is only dereferenced
 * during apk_prove calls, which the cgo pointer rules allow.  APK_ERR_HIP without a device (no CPU fallback: nothing to feed). */
int apk_host_alloc(int device, size_t bytes, void** ptr);
int apk_host_free(void* ptr);
int apk_host_register(void* ptr, size_t bytes);
int apk_host_unregister(void* ptr);

/* ---- device memory helpers for callers that keep inputs resident (bench, batched proofs) ------------------ */
int apk_device_alloc(apk_ctx* ctx, size_t bytes, void** d_ptr);
int apk_device_free(apk_ctx* ctx, void* d_ptr);
int apk_device_upload(apk_ctx* ctx, void* d_dst, const void* src, size_t bytes);
int apk_device_download(apk_ctx* ctx, void* dst, const void* d_src, size_t bytes);
int apk_device_copy(apk_ctx* ctx, void* d_dst, const void* d_src, size_t bytes);   /* device to device, same GPU */

/* ---- timing hooks used by bench.py: HIP-event time of the dominant kernel on the stream it runs on ------- */
typedef struct {
    double msm_accumulate_ms;  /* sum of msm_accumulate_kernel durations since the last reset */
    uint64_t msm_accumulate_launches;
    uint64_t msm_pairs;        /* (scalar, point) pairs those launches covered */
    double msm_total_ms;       /* whole MSM batches (digits .. final) */
    uint64_t msm_batches;
    double ntt_ms;             /* all NTT passes */
    uint64_t ntt_elements;     /* elements transformed (sum of sizes) */
    double prove_ms;           /* wall time inside apk_prove* */
    uint64_t proofs;
    /* host wall clock per round of the prover, summed over `proofs` (the rounds of SURVEY.md section 3.3: R1 = wire polynomials,
     * BSB22 commitments, [L][R][O]; R2 = grand product, [Z]; R3 = quotient, [H1..3]; R4 = evaluations, [lin], the two
     * openings).  Each includes the Fiat-Shamir hashing that follows it.  host_lincomb_ms = the part of R4 spent in the
     * host-side combination of commitments that replaces the MSM of the linearised polynomial. */
    double round_ms[4];
    double host_lincomb_ms;
    double msm_sort_ms;        /* of msm_total_ms: recoding + counting sort + scans (everything in front of msm_accumulate_kernel) */
    double msm_tail_ms;        /* of msm_total_ms: merge of unit partials, row/column sums, bit sums, final scaling (everything behind it) */
} apk_stats;
int apk_stats_enable(apk_ctx* ctx, int enable); /* enabling inserts hipEvents around the kernels above */
int apk_stats_read(apk_ctx* ctx, apk_stats* out, int reset);

/* ---- which forms the library took ------------------------------------------------------------------------------------------
 * Several kernels exist in two forms and the library picks one per launch from the LOAD it sees (how many proving slots are busy):
 * a lone proof gets the short-chain / latency forms, proofs under load the instruction-lean ones.  All forms give the same bytes;
 * the parity tests prove it by asserting, through these counters, that the loaded forms really ran while the blobs they compare
 * were produced (tests/test_gpu_parity.py::test_proofs_under_load_*).  Counted always, independent of apk_stats_enable - the
 * statistics synchronise the host with every batch and would change the very load the choices depend on. */
typedef struct {
    uint64_t proofs;                    /* apk_prove* calls that returned a proof */
    uint64_t msm_batches;               /* MSM launch sequences (1..4 MSMs each) */
    uint64_t msm_sort_two_level;        /* ... sorted in two levels (partitions, then LDS tiles) */
    uint64_t msm_sort_two_level_by_load;/* ... of those, two-level only BECAUSE other proofs were in flight (below 2^16 bases) */
    uint64_t msm_sort_fused;            /* ... two-level in the two-launch form */
    uint64_t msm_lean_tail;             /* lean tail: one lane per bucket in the merge from 32 768 buckets, lean row/column sums */
    uint64_t msm_rowcol_serial;         /* sixteen-lane serial row/column sums */
    uint64_t msm_combine_quad;          /* four lanes per addition in the merge (small lone batch) */
    uint64_t msm_small_units;           /* shrunken accumulate units (small lone batch) */
    uint64_t msm_one_launch;            /* whole MSM in one cooperative launch (small lone batch) */
    uint64_t msm_lagrange_wires;        /* [L][R][O] committed over the Lagrange SRS (small-scalar fast path) */
    uint64_t ntt_sequences;             /* NTT launch sequences */
    uint64_t ntt_radix4;                /* ... with radix-4 steps */
    uint64_t ntt_radix4_by_load;        /* ... radix-4 only BECAUSE other proofs were in flight (2^17..2^19) */
    uint64_t tail_fill_proofs;          /* lone proofs whose coset transforms ran beside the MSM tails (side stream) */
    uint64_t host_lincomb_pooled;       /* [lin] combinations of nearly idle contexts dealt to parked host threads (BLS12-381 by default; BN254 with APK_HOST_LINCOMB_THREADS > 1) */
    uint64_t msm_units_by_load;         /* MSM batches whose accumulate units were lengthened BECAUSE other proofs were in flight */
    uint64_t host_inputs;               /* proofs whose L, R, O came from host memory through a staging set (apk_prove) */
    uint64_t gang_proofs;               /* proofs made as members of a gang (two to four proofs sharing one stream and its MSM / NTT launches) */
    uint64_t gang_msm_launches;         /* MSM launch sequences that carried the batches of more than one proof */
    uint64_t gang_ntt_launches;         /* NTT launch sequences that carried the transforms of more than one proof */
    uint64_t gang_kernel_launches;      /* other kernels (grand product, quotient, evaluations, openings ...) launched once for several proofs */
    uint64_t reserved[2];
} apk_path_counts;
int apk_paths_read(apk_ctx* ctx, apk_path_counts* out, int reset);

#ifdef __cplusplus
}
#endif
#endif /* APK_H */
