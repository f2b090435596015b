/* ORACLE - TEST INFRASTRUCTURE, NOT PRODUCT CODE.  #included at the end of apk_oracle.c (one translation unit: it uses that file's
 * field / curve instantiations, SHA-256, codecs and domain helpers).
 *
 * A PERFORMANCE-FIRST host prover beside the clarity-first orc_prove: the same proof, byte for byte
 * (tests/test_oracle_c.py holds it to orc_prove and to oracle/plonk.py), written the way a CPU wants it so that bench.py's
 * cpu_baseline ("kind": "port") is a credible stand-in for gnark's CPU prover, which cannot be built here (no Go toolchain;
 * /root/reference/go.mod:8-9 pins gnark v0.15.0 / gnark-crypto v0.20.1, not vendored):
 *   - everything that depends on the circuit only is computed ONCE (orc_fast_setup): trace polynomials in canonical form and on the
 *     4n coset, the verifying-key commitments, twiddle tables - gnark keeps the equivalent in its ProvingKey / trace;
 *   - batch-affine Pippenger with signed digits and a cost-model window on a persistent thread pool (fast_msm_tmpl.h);
 *   - FFTs split over the pool: independent blocks for the short strides, butterfly ranges for the long ones;
 *   - the GPU path's schedule where it is plainly cheaper on any machine: quotient on ONE 4n coset, [lin] as a combination of
 *     commitments already in hand, lin(zeta) from the verifier's identity.
 * What it is not: assembly field arithmetic (gnark-crypto's mul is hand-written ADX/BMI2 assembly, ~1.5-2x this C), nor gnark's
 * exact task graph.  The number it produces is reported as a PORT, never as gnark.
 */
#include <stdatomic.h>
#include <malloc.h>
static void* fp_scratch(size_t need);
static void fp_scratch_release(void);

/* ---- persistent pool: run(fn, count) hands out task indices from an atomic counter; the caller takes part ---------------------- */
typedef struct fp_pool {
    int nthreads;                 /* workers besides the caller */
    pthread_t* th;
    pthread_mutex_t mu;
    pthread_cond_t wake, done;
    task_fn fn; void* arg; int count;
    atomic_int next;
    int active, gen, stop;
} fp_pool;
static void fp_drain(fp_pool* P) {
    for (;;) {
        const int i = atomic_fetch_add(&P->next, 1);
        if (i >= P->count) return;
        P->fn(P->arg, i);
    }
}
static void* fp_worker(void* a) {
    fp_pool* P = (fp_pool*)a;
    int seen = 0;
    pthread_mutex_lock(&P->mu);
    for (;;) {
        while (!P->stop && P->gen == seen) pthread_cond_wait(&P->wake, &P->mu);
        if (P->stop) { pthread_mutex_unlock(&P->mu); fp_scratch_release(); return NULL; }
        seen = P->gen;
        pthread_mutex_unlock(&P->mu);
        fp_drain(P);
        pthread_mutex_lock(&P->mu);
        if (--P->active == 0) pthread_cond_signal(&P->done);
    }
}
static fp_pool* fp_pool_create(int threads) {
    fp_pool* P = (fp_pool*)calloc(1, sizeof *P);
    P->nthreads = threads > 1 ? threads - 1 : 0;
    if (P->nthreads > 511) P->nthreads = 511;
    pthread_mutex_init(&P->mu, NULL); pthread_cond_init(&P->wake, NULL); pthread_cond_init(&P->done, NULL);
    P->th = (pthread_t*)calloc((size_t)P->nthreads + 1, sizeof(pthread_t));
    for (int i = 0; i < P->nthreads; i++) pthread_create(&P->th[i], NULL, fp_worker, P);
    return P;
}
static void fp_pool_destroy(fp_pool* P) {
    pthread_mutex_lock(&P->mu); P->stop = 1; pthread_cond_broadcast(&P->wake); pthread_mutex_unlock(&P->mu);
    for (int i = 0; i < P->nthreads; i++) pthread_join(P->th[i], NULL);
    free(P->th); free(P);
}
static void fp_pool_run(fp_pool* P, task_fn fn, void* arg, int count) {
    if (count <= 0) return;
    if (P->nthreads == 0 || count == 1) { for (int i = 0; i < count; i++) fn(arg, i); return; }
    pthread_mutex_lock(&P->mu);
    P->fn = fn; P->arg = arg; P->count = count; atomic_store(&P->next, 0);
    P->active = P->nthreads; P->gen++;
    pthread_cond_broadcast(&P->wake);
    pthread_mutex_unlock(&P->mu);
    fp_drain(P);
    pthread_mutex_lock(&P->mu);
    while (P->active) pthread_cond_wait(&P->done, &P->mu);
    pthread_mutex_unlock(&P->mu);
}

/* per-thread scratch for the tasks' working sets: grows, is reused, is freed when a pool worker exits (the calling thread keeps
 * its own for its lifetime) */
static __thread void* fp_tls_mem;
static __thread size_t fp_tls_cap;
static void* fp_scratch(size_t need) {
    if (fp_tls_cap < need) {
        free(fp_tls_mem);
        fp_tls_cap = need + need / 4 + 4096;
        fp_tls_mem = malloc(fp_tls_cap);
    }
    return fp_tls_mem;
}
static void fp_scratch_release(void) { free(fp_tls_mem); fp_tls_mem = NULL; fp_tls_cap = 0; }

/* ---- the fast MSM for both curves ------------------------------------------------------------------------------------------- */
#define FPN(x) f4_##x
#define CN(x) bn_##x
#include "fast_msm_tmpl.h"
#undef FPN
#undef CN
#define FPN(x) f6_##x
#define CN(x) bls_##x
#include "fast_msm_tmpl.h"
#undef FPN
#undef CN

typedef struct { const fr_field* F; const fr_t* in; uint64_t* out; size_t n, per; } fp_plain_job;
static void fp_plain_task(void* a, int t) {
    fp_plain_job* J = (fp_plain_job*)a;
    const size_t lo = (size_t)t * J->per, hi = lo + J->per < J->n ? lo + J->per : J->n;
    for (size_t i = lo; i < hi; i++) { fr_t c; f4_from_mont(J->F, &c, &J->in[i]); memcpy(J->out + 4 * i, c.l, 32); }
}
static void fp_commit(int curve, const void* srs, const fr_t* coeffs, size_t n, fp_pool* pool, int threads, void* out_aff) {
    uint64_t* plain = (uint64_t*)malloc(n * 32 + 32);
    fp_plain_job J = {&FR[curve], coeffs, plain, n, (n + (size_t)threads * 4 - 1) / ((size_t)threads * 4)};
    if (J.per < 1) J.per = 1;
    fp_pool_run(pool, fp_plain_task, &J, (int)((n + J.per - 1) / J.per));
    if (curve == 0) bn_fmsm(&FP_BN, (const bn_aff*)srs, plain, n, SCALAR_BITS[0], pool, threads, (bn_aff*)out_aff);
    else bls_fmsm(&FP_BLS, (const bls_aff*)srs, plain, n, SCALAR_BITS[1], pool, threads, (bls_aff*)out_aff);
    free(plain);
}
int orc_msm_fast(int curve, const void* points, const void* scalars, uint64_t n, int threads, void* out) {
    orc_init();
    if (curve != 0 && curve != 1) return 1;
    fp_pool* pool = fp_pool_create(threads);
    fp_commit(curve, points, (const fr_t*)scalars, n, pool, threads < 1 ? 1 : threads, out);
    fp_pool_destroy(pool);
    return 0;
}

/* ---- parallel FFT: bit reversal, then the stages; short strides as independent blocks, long strides by butterfly range ------- */
typedef struct { const fr_field* F; fr_t* a; const fr_t* w; size_t n; int lg; size_t blk; size_t len; size_t per; } fp_fft_job;
static void fp_bitrev_task(void* arg, int t) {
    fp_fft_job* J = (fp_fft_job*)arg;
    const size_t lo = (size_t)t * J->per, hi = lo + J->per < J->n ? lo + J->per : J->n;
    for (size_t i = lo; i < hi; i++) {
        size_t j = 0;
        for (int b = 0; b < J->lg; b++) j |= ((i >> b) & 1) << (J->lg - 1 - b);
        if (i < j) { fr_t x = J->a[i]; J->a[i] = J->a[j]; J->a[j] = x; }
    }
}
/* all stages with len <= blk inside one block of blk elements */
static void fp_fft_block_task(void* arg, int t) {
    fp_fft_job* J = (fp_fft_job*)arg;
    const fr_field* F = J->F;
    fr_t* a = J->a + (size_t)t * J->blk;
    for (size_t len = 2; len <= J->blk; len <<= 1) {
        const size_t half = len >> 1, step = J->n / len;
        for (size_t s = 0; s < J->blk; s += len)
            for (size_t j = 0; j < half; j++) {
                fr_t u = a[s + j], v;
                f4_mul(F, &v, &a[s + j + half], &J->w[j * step]);
                f4_add(F, &a[s + j], &u, &v);
                f4_sub(F, &a[s + j + half], &u, &v);
            }
    }
}
/* one long stage: butterflies [lo, hi) of the n/2 of that stage */
static void fp_fft_stage_task(void* arg, int t) {
    fp_fft_job* J = (fp_fft_job*)arg;
    const fr_field* F = J->F;
    const size_t half = J->len >> 1, step = J->n / J->len, total = J->n >> 1;
    const size_t lo = (size_t)t * J->per, hi = lo + J->per < total ? lo + J->per : total;
    for (size_t k = lo; k < hi; k++) {
        const size_t s = (k / half) * J->len, j = k % half;
        fr_t u = J->a[s + j], v;
        f4_mul(F, &v, &J->a[s + j + half], &J->w[j * step]);
        f4_add(F, &J->a[s + j], &u, &v);
        f4_sub(F, &J->a[s + j + half], &u, &v);
    }
}
static void fp_fft(int curve, fr_t* a, size_t n, const fr_t* w, fp_pool* pool, int threads) {
    int lg = 0; while (((size_t)1 << lg) < n) lg++;
    fp_fft_job J = {&FR[curve], a, w, n, lg, 0, 0, 0};
    const size_t parts = (size_t)threads * 4;
    J.per = (n + parts - 1) / parts; if (J.per < 64) J.per = 64;
    fp_pool_run(pool, fp_bitrev_task, &J, (int)((n + J.per - 1) / J.per));
    size_t blk = n;
    while (blk > 1024 && n / blk < parts) blk >>= 1;      /* at least `parts` blocks, blocks of >= 1024 elements (cache resident) */
    J.blk = blk;
    fp_pool_run(pool, fp_fft_block_task, &J, (int)(n / blk));
    for (size_t len = blk << 1; len <= n; len <<= 1) {
        J.len = len;
        J.per = ((n >> 1) + parts - 1) / parts; if (J.per < 256) J.per = 256;
        fp_pool_run(pool, fp_fft_stage_task, &J, (int)(((n >> 1) + J.per - 1) / J.per));
    }
}
typedef struct { const fr_field* F; fr_t* a; const fr_t* tab; fr_t k; size_t n, per; int mode; } fp_scale_job;
static void fp_scale_task(void* arg, int t) {      /* mode 0: a[i] *= k;  1: a[i] *= tab[i];  2: a[i] *= tab[i] * k */
    fp_scale_job* J = (fp_scale_job*)arg;
    const size_t lo = (size_t)t * J->per, hi = lo + J->per < J->n ? lo + J->per : J->n;
    for (size_t i = lo; i < hi; i++) {
        if (J->mode != 0) f4_mul(J->F, &J->a[i], &J->a[i], &J->tab[i]);
        if (J->mode != 1) f4_mul(J->F, &J->a[i], &J->a[i], &J->k);
    }
}
static void fp_scale(int curve, fr_t* a, size_t n, const fr_t* tab, const fr_t* k, fp_pool* pool, int threads) {
    fp_scale_job J; memset(&J, 0, sizeof J); J.F = &FR[curve]; J.a = a; J.tab = tab; J.n = n; J.mode = tab ? (k ? 2 : 1) : 0;
    if (k) J.k = *k;
    J.per = (n + (size_t)threads * 4 - 1) / ((size_t)threads * 4); if (J.per < 256) J.per = 256;
    fp_pool_run(pool, fp_scale_task, &J, (int)((n + J.per - 1) / J.per));
}

#define FP_MAX_INJ 10
/* orc_circuit + the BSB22 part of gnark's trace and proving key */
typedef struct {
    orc_circuit base;
    uint32_t nb_commit;                /* <= 2 */
    uint32_t cci[2];                   /* VK CommitmentConstraintIndexes */
    const void* qcp[2];                /* Lagrange, n Fr each */
    const void* srs_lagrange;          /* n G1 affine */
} orc_circuit_ex;

/* gnark fr.Hash(msg, "BSB22-Plonk", 1) = expand_msg_xmd(sha256, 48 bytes) mod r, as the verifier recomputes it
 * (templateLogicSigBN254.go:386-397) */
static void fp_hash_fr(const fr_field* F, fr_t* out, const uint8_t* msg, size_t len) {
    static const uint8_t dst_prime[12] = {'B', 'S', 'B', '2', '2', '-', 'P', 'l', 'o', 'n', 'k', 0x0b};
    const uint8_t zeros[64] = {0}, lib[3] = {0x00, 0x30, 0x00}, one = 1, two = 2;
    uint8_t b0[32], b1[32], b2[32], x[32], lo[32] = {0};
    sha_t h;
    sha_init(&h); sha_update(&h, zeros, 64); sha_update(&h, msg, len); sha_update(&h, lib, 3); sha_update(&h, dst_prime, 12); sha_final(&h, b0);
    sha_init(&h); sha_update(&h, b0, 32); sha_update(&h, &one, 1); sha_update(&h, dst_prime, 12); sha_final(&h, b1);
    for (int i = 0; i < 32; i++) x[i] = b0[i] ^ b1[i];
    sha_init(&h); sha_update(&h, x, 32); sha_update(&h, &two, 1); sha_update(&h, dst_prime, 12); sha_final(&h, b2);
    memcpy(lo + 16, b2, 16);
    fr_t hi, lw, t128, t; memset(&t, 0, sizeof t); t.l[2] = 1;       /* 2^128 */
    f4_to_mont(F, &t128, &t);
    fr_from_be_reduce(F, &hi, b1); fr_from_be_reduce(F, &lw, lo);
    f4_mul(F, &hi, &hi, &t128);
    f4_add(F, out, &hi, &lw);
}

/* ---- the context: what depends on the circuit only ---------------------------------------------------------------------------- */
enum { FQL, FQR, FQM, FQO, FQK, FS1, FS2, FS3, FNTRACE };
typedef struct fp_ctx {
    int curve; size_t n; uint32_t nb_public;
    void* srs;                     /* n + 3 points (copied) */
    fr_t *w0, *w0i, *w1, *w1i;     /* omega^i, omega^-i of the n and the 4n domain (half tables) */
    fr_t ninv, n4inv, omega, u;
    fr_t* upow;                    /* u^i, i < n + 3 */
    fr_t* uinv_pow;                /* u^-i / (4n), i < 4n */
    fr_t* omega_pow;               /* omega^i, i < n */
    fr_t* tl[FNTRACE];             /* Lagrange (only S1..S3 and Qk are read by the prover) */
    fr_t* tc[FNTRACE];             /* canonical */
    fr_t* te[FNTRACE];             /* on the 4n coset (Qk: trace only, public rows zero) */
    fr_t* l0e;                     /* L_0 on the coset */
    /* rows of Qk a proof writes: the public inputs 0 .. nb_public-1, then nb_public + cci_k for the BSB22 commitments; their
     * Lagrange polynomials on the coset (row 0 = l0e) complete Qk inside the quotient loop */
    int n_inj; uint32_t inj_row[FP_MAX_INJ]; fr_t* inj_e[FP_MAX_INJ];
    fr_t zhinv[4];
    uint8_t vk_pt[FNTRACE][96], vkb[FNTRACE][96];
    /* BSB22 (bsb22_test.go:18-39; templateLogicSigBN254.go:386-397): Qcp_k canonical / on the coset / committed, the Lagrange SRS */
    uint32_t nb_commit, cci[2];
    void* srs_lag;
    fr_t *qcp_c[2], *qcp_e[2];
    uint8_t qcp_pt[2][96], qcp_b[2][96];
} fp_ctx;

static fr_t* fp_alloc(size_t n) { return (fr_t*)calloc(n, sizeof(fr_t)); }
/* canonical polynomial (len coefficients) -> its 4n evaluations on the coset u<w_4n>, into out (4n) */
static void fp_coset_eval(const fp_ctx* X, const fr_t* can, size_t len, fr_t* out, fp_pool* pool, int threads) {
    const size_t n4 = 4 * X->n;
    memcpy(out, can, len * sizeof(fr_t));
    memset(out + len, 0, (n4 - len) * sizeof(fr_t));
    fp_scale(X->curve, out, len, X->upow, NULL, pool, threads);
    fp_fft(X->curve, out, n4, X->w1, pool, threads);
}
static void fp_ifft_n(const fp_ctx* X, fr_t* a, fp_pool* pool, int threads) {
    fp_fft(X->curve, a, X->n, X->w0i, pool, threads);
    fp_scale(X->curve, a, X->n, NULL, &X->ninv, pool, threads);
}

void orc_fast_free(fp_ctx* X) {
    if (!X) return;
    free(X->srs); free(X->w0); free(X->w0i); free(X->w1); free(X->w1i); free(X->upow); free(X->uinv_pow); free(X->omega_pow);
    for (int i = 0; i < FNTRACE; i++) { free(X->tl[i]); free(X->tc[i]); free(X->te[i]); }
    for (int i = 0; i < X->n_inj; i++) if (X->inj_e[i] != X->l0e) free(X->inj_e[i]);
    free(X->l0e); free(X->srs_lag);
    for (int k = 0; k < 2; k++) { free(X->qcp_c[k]); free(X->qcp_e[k]); }
    free(X);
}

int orc_fast_setup_ex(const orc_circuit_ex* E, int threads, fp_ctx** out);
int orc_fast_setup(const orc_circuit* C, int threads, fp_ctx** out) {
    orc_circuit_ex E; memset(&E, 0, sizeof E); E.base = *C;
    return orc_fast_setup_ex(&E, threads, out);
}
int orc_fast_setup_ex(const orc_circuit_ex* E, int threads, fp_ctx** out) {
    const orc_circuit* C = &E->base;
    if (E->nb_commit > 2 || (E->nb_commit && !E->srs_lagrange) || C->nb_public + E->nb_commit > FP_MAX_INJ) return 2;
    orc_init();
    {   /* many threads of ONE process allocate and free multi-megabyte buffers per proof: keep them inside malloc's arenas instead
         * of an mmap / munmap pair each (the address-space lock serialises those across all threads) */
        static int tuned = 0;
        if (!tuned) { tuned = 1; mallopt(M_MMAP_THRESHOLD, 32 << 20); mallopt(M_TRIM_THRESHOLD, 1 << 30); mallopt(M_ARENA_MAX, 256);   /* 32 MiB = glibc's cap */ }
    }
    const int cv = C->curve;
    if (cv != 0 && cv != 1) return 1;
    if (threads < 1) threads = 1;
    const fr_field* F = &FR[cv];
    fp_ctx* X = (fp_ctx*)calloc(1, sizeof *X);
    const size_t n = C->n, n4 = 4 * n, PT = g1_size(cv);
    X->curve = cv; X->n = n; X->nb_public = C->nb_public;
    X->srs = malloc((n + 3) * PT); memcpy(X->srs, C->srs, (n + 3) * PT);
    fp_pool* pool = fp_pool_create(threads);
    {
        domain_t d0, d1; domain_init(cv, &d0, n); domain_init(cv, &d1, n4);
        X->w0 = d0.w; X->w0i = d0.wi; X->w1 = d1.w; X->w1i = d1.wi;
        X->ninv = d0.ninv; X->n4inv = d1.ninv; X->omega = d0.n > 1 ? d0.w[1] : F->one;
    }
    fr_set_u64(F, &X->u, SHIFT[cv]);
    X->upow = fp_alloc(n + 3); X->uinv_pow = fp_alloc(n4); X->omega_pow = fp_alloc(n);
    X->upow[0] = F->one; for (size_t i = 1; i < n + 3; i++) f4_mul(F, &X->upow[i], &X->upow[i - 1], &X->u);
    { fr_t ui; f4_inv(F, &ui, &X->u); X->uinv_pow[0] = X->n4inv; for (size_t i = 1; i < n4; i++) f4_mul(F, &X->uinv_pow[i], &X->uinv_pow[i - 1], &ui); }
    X->omega_pow[0] = F->one; for (size_t i = 1; i < n; i++) f4_mul(F, &X->omega_pow[i], &X->omega_pow[i - 1], &X->omega);
    fr_t u2; f4_sqr(F, &u2, &X->u);
    const void* cols[5] = {C->ql, C->qr, C->qm, C->qo, C->qk};
    for (int i = 0; i < FNTRACE; i++) { X->tl[i] = fp_alloc(n); X->tc[i] = fp_alloc(n); X->te[i] = fp_alloc(n4); }
    for (int i = 0; i < 5; i++) memcpy(X->tl[i], cols[i], n * sizeof(fr_t));
    for (int j = 0; j < 3; j++)
        for (size_t i = 0; i < n; i++) {
            const int64_t p = C->perm[(size_t)j * n + i];
            const size_t blk = (size_t)p / n, pos = (size_t)p % n;
            X->tl[FS1 + j][i] = X->omega_pow[pos];
            if (blk == 1) f4_mul(F, &X->tl[FS1 + j][i], &X->tl[FS1 + j][i], &X->u);
            if (blk == 2) f4_mul(F, &X->tl[FS1 + j][i], &X->tl[FS1 + j][i], &u2);
        }
    for (int i = 0; i < FNTRACE; i++) {
        memcpy(X->tc[i], X->tl[i], n * sizeof(fr_t));
        fp_ifft_n(X, X->tc[i], pool, threads);
        fp_coset_eval(X, X->tc[i], n, X->te[i], pool, threads);
        fp_commit(cv, X->srs, X->tc[i], n, pool, threads, X->vk_pt[i]);
        g1_raw(cv, X->vk_pt[i], X->vkb[i]);
    }
    /* BSB22: Qcp_k canonical, on the coset, committed; the Lagrange SRS */
    X->nb_commit = E->nb_commit;
    if (E->nb_commit) { X->srs_lag = malloc(n * PT); memcpy(X->srs_lag, E->srs_lagrange, n * PT); }
    for (uint32_t k = 0; k < E->nb_commit; k++) {
        X->cci[k] = E->cci[k];
        X->qcp_c[k] = fp_alloc(n); X->qcp_e[k] = fp_alloc(n4);
        memcpy(X->qcp_c[k], E->qcp[k], n * sizeof(fr_t));
        fp_ifft_n(X, X->qcp_c[k], pool, threads);
        fp_coset_eval(X, X->qcp_c[k], n, X->qcp_e[k], pool, threads);
        fp_commit(cv, X->srs, X->qcp_c[k], n, pool, threads, X->qcp_pt[k]);
        g1_raw(cv, X->qcp_pt[k], X->qcp_b[k]);
    }
    /* L_0 and the Lagrange polynomials of the written rows on the coset: L_r(X) = (1/n) sum_i omega^(-r i) X^i */
    fr_t* tmp = fp_alloc(n);
    X->l0e = fp_alloc(n4);
    for (size_t i = 0; i < n; i++) tmp[i] = X->ninv;
    fp_coset_eval(X, tmp, n, X->l0e, pool, threads);
    X->n_inj = 0;
    for (uint32_t r = 0; r < C->nb_public; r++) X->inj_row[X->n_inj++] = r;
    for (uint32_t k = 0; k < E->nb_commit; k++) X->inj_row[X->n_inj++] = C->nb_public + E->cci[k];
    for (int j = 0; j < X->n_inj; j++) {
        const uint32_t r = X->inj_row[j];
        if (r >= n) { free(tmp); fp_pool_destroy(pool); orc_fast_free(X); return 2; }
        if (r == 0) { X->inj_e[j] = X->l0e; continue; }
        fr_t wr, cur = X->ninv; f4_inv(F, &wr, &X->omega_pow[r]);
        for (size_t i = 0; i < n; i++) { tmp[i] = cur; f4_mul(F, &cur, &cur, &wr); }
        X->inj_e[j] = fp_alloc(n4);
        fp_coset_eval(X, tmp, n, X->inj_e[j], pool, threads);
    }
    free(tmp);
    {
        const fr_t w4 = X->w1[1];
        fr_t un, i4, cur; f4_pow_u64(F, &un, &X->u, n); f4_pow_u64(F, &i4, &w4, n); cur = un;
        for (int k = 0; k < 4; k++) { fr_t t; f4_sub(F, &t, &cur, &F->one); f4_inv(F, &X->zhinv[k], &t); f4_mul(F, &cur, &cur, &i4); }
    }
    fp_pool_destroy(pool);
    *out = X;
    return 0;
}

/* ---- per-proof pieces ----------------------------------------------------------------------------------------------------------- */
typedef struct {
    const fp_ctx* X; const fr_t *L, *R, *O; fr_t beta, gamma, bu, bu2; fr_t *num, *den; size_t per;
} fp_gp_job;
static void fp_gp_terms_task(void* arg, int t) {
    fp_gp_job* J = (fp_gp_job*)arg;
    const fp_ctx* X = J->X; const fr_field* F = &FR[X->curve];
    const size_t lo = (size_t)t * J->per, hi = lo + J->per < X->n ? lo + J->per : X->n;
    for (size_t i = lo; i < hi; i++) {
        fr_t l, r, o, tt, a, b, c;
        f4_add(F, &l, &J->L[i], &J->gamma); f4_add(F, &r, &J->R[i], &J->gamma); f4_add(F, &o, &J->O[i], &J->gamma);
        f4_mul(F, &tt, &J->beta, &X->omega_pow[i]); f4_add(F, &a, &l, &tt);
        f4_mul(F, &tt, &J->bu, &X->omega_pow[i]); f4_add(F, &b, &r, &tt);
        f4_mul(F, &tt, &J->bu2, &X->omega_pow[i]); f4_add(F, &c, &o, &tt);
        f4_mul(F, &J->num[i], &a, &b); f4_mul(F, &J->num[i], &J->num[i], &c);
        f4_mul(F, &tt, &J->beta, &X->tl[FS1][i]); f4_add(F, &a, &l, &tt);
        f4_mul(F, &tt, &J->beta, &X->tl[FS2][i]); f4_add(F, &b, &r, &tt);
        f4_mul(F, &tt, &J->beta, &X->tl[FS3][i]); f4_add(F, &c, &o, &tt);
        f4_mul(F, &J->den[i], &a, &b); f4_mul(F, &J->den[i], &J->den[i], &c);
    }
}

typedef struct {
    const fp_ctx* X; fr_t *el, *er, *eo, *ez, *h; fr_t alpha, a2, beta, gamma, bu, bu2; fr_t delta[FP_MAX_INJ]; fr_t* epi2[2]; size_t per;
} fp_quot_job;
static void fp_quot_task(void* arg, int t) {
    fp_quot_job* Q = (fp_quot_job*)arg;
    const fp_ctx* X = Q->X; const fr_field* F = &FR[X->curve];
    const size_t n4 = 4 * X->n, lo = (size_t)t * Q->per, hi = lo + Q->per < n4 ? lo + Q->per : n4;
    const fr_t w4 = X->w1[1];
    fr_t x; f4_pow_u64(F, &x, &w4, lo); f4_mul(F, &x, &x, &X->u);
    for (size_t i = lo; i < hi; i++) {
        const fr_t l = Q->el[i], r = Q->er[i], o = Q->eo[i], z = Q->ez[i], zs = Q->ez[(i + 4) % n4];
        fr_t gate, tt, lg, rg, og, pa, pb, a, b, c, loc, num, qk = X->te[FQK][i];
        /* completed Qk = trace Qk + sum_r (pub_r - trace Qk[r]) L_r */
        for (int rr = 0; rr < X->n_inj; rr++) { f4_mul(F, &tt, &Q->delta[rr], &X->inj_e[rr][i]); f4_add(F, &qk, &qk, &tt); }
        f4_mul(F, &gate, &X->te[FQL][i], &l);
        f4_mul(F, &tt, &X->te[FQR][i], &r); f4_add(F, &gate, &gate, &tt);
        f4_mul(F, &tt, &l, &r); f4_mul(F, &tt, &tt, &X->te[FQM][i]); f4_add(F, &gate, &gate, &tt);
        f4_mul(F, &tt, &X->te[FQO][i], &o); f4_add(F, &gate, &gate, &tt);
        f4_add(F, &gate, &gate, &qk);
        for (uint32_t k = 0; k < X->nb_commit; k++) { f4_mul(F, &tt, &X->qcp_e[k][i], &Q->epi2[k][i]); f4_add(F, &gate, &gate, &tt); }
        f4_add(F, &lg, &l, &Q->gamma); f4_add(F, &rg, &r, &Q->gamma); f4_add(F, &og, &o, &Q->gamma);
        f4_mul(F, &tt, &Q->beta, &X->te[FS1][i]); f4_add(F, &a, &lg, &tt);
        f4_mul(F, &tt, &Q->beta, &X->te[FS2][i]); f4_add(F, &b, &rg, &tt);
        f4_mul(F, &tt, &Q->beta, &X->te[FS3][i]); f4_add(F, &c, &og, &tt);
        f4_mul(F, &pa, &zs, &a); f4_mul(F, &pa, &pa, &b); f4_mul(F, &pa, &pa, &c);
        f4_mul(F, &tt, &Q->beta, &x); f4_add(F, &a, &lg, &tt);
        f4_mul(F, &tt, &Q->bu, &x); f4_add(F, &b, &rg, &tt);
        f4_mul(F, &tt, &Q->bu2, &x); f4_add(F, &c, &og, &tt);
        f4_mul(F, &pb, &z, &a); f4_mul(F, &pb, &pb, &b); f4_mul(F, &pb, &pb, &c);
        f4_sub(F, &tt, &z, &F->one); f4_mul(F, &loc, &X->l0e[i], &tt);
        f4_sub(F, &tt, &pa, &pb); f4_mul(F, &tt, &tt, &Q->alpha); f4_add(F, &num, &gate, &tt);
        f4_mul(F, &tt, &Q->a2, &loc); f4_add(F, &num, &num, &tt);
        f4_mul(F, &Q->h[i], &num, &X->zhinv[i & 3]);
        f4_mul(F, &x, &x, &w4);
    }
}

/* f(x) by chunks: partial[t] = sum_{i in chunk} c_i x^i */
typedef struct { const fr_field* F; const fr_t* c; size_t len, per; fr_t x; fr_t* partial; } fp_eval_job;
static void fp_eval_task(void* arg, int t) {
    fp_eval_job* J = (fp_eval_job*)arg;
    const size_t lo = (size_t)t * J->per, hi = lo + J->per < J->len ? lo + J->per : J->len;
    fr_t acc; memset(&acc, 0, sizeof acc);
    for (size_t i = hi; i-- > lo;) { f4_mul(J->F, &acc, &acc, &J->x); f4_add(J->F, &acc, &acc, &J->c[i]); }
    fr_t xp; f4_pow_u64(J->F, &xp, &J->x, lo);
    f4_mul(J->F, &J->partial[t], &acc, &xp);
}
static void fp_poly_eval(int curve, fr_t* r, const fr_t* c, size_t len, const fr_t* x, fp_pool* pool, int threads) {
    const fr_field* F = &FR[curve];
    fp_eval_job J; J.F = F; J.c = c; J.len = len; J.x = *x;
    J.per = (len + (size_t)threads * 2 - 1) / ((size_t)threads * 2); if (J.per < 512) J.per = 512;
    const int parts = (int)((len + J.per - 1) / J.per);
    J.partial = fp_alloc((size_t)parts);
    fp_pool_run(pool, fp_eval_task, &J, parts);
    fr_t acc; memset(&acc, 0, sizeof acc);
    for (int i = 0; i < parts; i++) f4_add(F, &acc, &acc, &J.partial[i]);
    free(J.partial);
    *r = acc;
}

/* sum_i k_i P_i for a handful of points (the [lin] combination): plain double-and-add, Jacobian */
static void fp_small_msm(int cv, uint8_t (*pts)[96], const fr_t* ks, int count, void* out_aff) {
    const fr_field* F = &FR[cv];
    if (cv == 0) {
        bn_jac acc; bn_jac_set_inf(&FP_BN, &acc);
        for (int i = 0; i < count; i++) {
            fr_t k; f4_from_mont(F, &k, &ks[i]);
            bn_jac r; bn_jac_set_inf(&FP_BN, &r);
            for (int b = 255; b >= 0; b--) { bn_jac_dbl(&FP_BN, &r, &r); if ((k.l[b >> 6] >> (b & 63)) & 1) bn_jac_madd(&FP_BN, &r, &r, (const bn_aff*)pts[i], 0); }
            bn_jac_add(&FP_BN, &acc, &acc, &r);
        }
        bn_jac_to_aff(&FP_BN, (bn_aff*)out_aff, &acc);
    } else {
        bls_jac acc; bls_jac_set_inf(&FP_BLS, &acc);
        for (int i = 0; i < count; i++) {
            fr_t k; f4_from_mont(F, &k, &ks[i]);
            bls_jac r; bls_jac_set_inf(&FP_BLS, &r);
            for (int b = 255; b >= 0; b--) { bls_jac_dbl(&FP_BLS, &r, &r); if ((k.l[b >> 6] >> (b & 63)) & 1) bls_jac_madd(&FP_BLS, &r, &r, (const bls_aff*)pts[i], 0); }
            bls_jac_add(&FP_BLS, &acc, &acc, &r);
        }
        bls_jac_to_aff(&FP_BLS, (bls_aff*)out_aff, &acc);
    }
}

typedef struct { const fr_field* F; fr_t* out; const fr_t* const* ps; const size_t* ls; const fr_t* ks; int count; size_t n, per; } fp_lc_job;
static void fp_lc_task(void* arg, int t) {       /* out[i] = sum_k ks[k] ps[k][i] */
    fp_lc_job* J = (fp_lc_job*)arg;
    const size_t lo = (size_t)t * J->per, hi = lo + J->per < J->n ? lo + J->per : J->n;
    for (size_t i = lo; i < hi; i++) {
        fr_t acc, tt; memset(&acc, 0, sizeof acc);
        for (int k = 0; k < J->count; k++) if (i < J->ls[k]) { f4_mul(J->F, &tt, &J->ks[k], &J->ps[k][i]); f4_add(J->F, &acc, &acc, &tt); }
        J->out[i] = acc;
    }
}

/* plonk.Prove (algoplonk.go:89) on a prepared context; the bytes of orc_prove (and, with commitments, of oracle/plonk.py).
 * pi2[k] = the BSB22 committed columns (Lagrange, n Fr each, hiding entries placed) */
int orc_fast_prove_ex(const fp_ctx* X, const void* Lp, const void* Rp, const void* Op, const void* pubp, const void* blindp,
                      const void* const* pi2p, int threads, uint8_t* blob, uint64_t* blob_len, uint8_t* challenges_out);
int orc_fast_prove(const fp_ctx* X, const void* Lp, const void* Rp, const void* Op, const void* pubp, const void* blindp, int threads,
                   uint8_t* blob, uint64_t* blob_len, uint8_t* challenges_out) {
    return orc_fast_prove_ex(X, Lp, Rp, Op, pubp, blindp, NULL, threads, blob, blob_len, challenges_out);
}
int orc_fast_prove_ex(const fp_ctx* X, const void* Lp, const void* Rp, const void* Op, const void* pubp, const void* blindp,
                      const void* const* pi2p, int threads, uint8_t* blob, uint64_t* blob_len, uint8_t* challenges_out) {
    const int cv = X->curve;
    const fr_field* F = &FR[cv];
    if (threads < 1) threads = 1;
    const uint32_t nbc = X->nb_commit;
    if (nbc && !pi2p) return 2;
    const size_t n = X->n, n4 = 4 * n, PT = g1_size(cv);
    const fr_t *L = (const fr_t*)Lp, *R = (const fr_t*)Rp, *O = (const fr_t*)Op, *pub = (const fr_t*)pubp, *bl = (const fr_t*)blindp;
    fp_pool* pool = fp_pool_create(threads);
    const fr_t u = X->u; fr_t u2; f4_sqr(F, &u2, &u);

    /* ---- round 1: BSB22 commitments (kzg.Commit over the Lagrange SRS, hash to a field element), wires ---- */
    uint8_t bsb_pt[2][96], bsb_b[2][96];
    fr_t cval[2];
    fr_t *pi2c[2] = {NULL, NULL}, *epi2[2] = {NULL, NULL};
    for (uint32_t k = 0; k < nbc; k++) {
        fp_commit(cv, X->srs_lag, (const fr_t*)pi2p[k], n, pool, threads, bsb_pt[k]);
        g1_raw(cv, bsb_pt[k], bsb_b[k]);
        fp_hash_fr(F, &cval[k], bsb_b[k], PT);
        pi2c[k] = fp_alloc(n); epi2[k] = fp_alloc(n4);
        memcpy(pi2c[k], pi2p[k], n * sizeof(fr_t));
        fp_ifft_n(X, pi2c[k], pool, threads);
        fp_coset_eval(X, pi2c[k], n, epi2[k], pool, threads);
    }
    fr_t* wc[4];
    for (int j = 0; j < 4; j++) wc[j] = fp_alloc(n + 3);
    memcpy(wc[0], L, n * sizeof(fr_t)); memcpy(wc[1], R, n * sizeof(fr_t)); memcpy(wc[2], O, n * sizeof(fr_t));
    for (int j = 0; j < 3; j++) fp_ifft_n(X, wc[j], pool, threads);
    for (int j = 0; j < 3; j++)
        for (int k = 0; k < 2; k++) { f4_sub(F, &wc[j][k], &wc[j][k], &bl[2 * j + k]); f4_add(F, &wc[j][n + k], &wc[j][n + k], &bl[2 * j + k]); }
    uint8_t lro_pt[3][96], lro_b[3][96];
    for (int j = 0; j < 3; j++) { fp_commit(cv, X->srs, wc[j], n + 2, pool, threads, lro_pt[j]); g1_raw(cv, lro_pt[j], lro_b[j]); }
    uint8_t* pub_b = (uint8_t*)malloc((size_t)X->nb_public * 32 + 1);
    for (uint32_t i = 0; i < X->nb_public; i++) fr_to_be(F, &pub[i], pub_b + 32 * i);
    uint8_t gamma_raw[32], beta_raw[32], alpha_raw[32], zeta_raw[32];
    {
        const uint8_t* parts[14]; size_t lens[14]; int np = 0;
        const uint8_t* head[8] = {X->vkb[FS1], X->vkb[FS2], X->vkb[FS3], X->vkb[FQL], X->vkb[FQR], X->vkb[FQM], X->vkb[FQO], X->vkb[FQK]};
        for (int i = 0; i < 8; i++) { parts[np] = head[i]; lens[np++] = PT; }
        for (uint32_t k = 0; k < nbc; k++) { parts[np] = X->qcp_b[k]; lens[np++] = PT; }
        parts[np] = pub_b; lens[np++] = (size_t)X->nb_public * 32;
        for (int j = 0; j < 3; j++) { parts[np] = lro_b[j]; lens[np++] = PT; }
        challenge("gamma", NULL, parts, lens, np, gamma_raw);
        challenge("beta", gamma_raw, NULL, NULL, 0, beta_raw);
    }
    fr_t gamma, beta; fr_from_be_reduce(F, &gamma, gamma_raw); fr_from_be_reduce(F, &beta, beta_raw);
    fr_t bu, bu2; f4_mul(F, &bu, &beta, &u); f4_mul(F, &bu2, &beta, &u2);

    /* ---- round 2: grand product (terms in parallel, the two scans serial: 2 n products) ---- */
    {
        fr_t *num = fp_alloc(n), *den = fp_alloc(n), *pre = fp_alloc(n);
        fp_gp_job G; G.X = X; G.L = L; G.R = R; G.O = O; G.beta = beta; G.gamma = gamma; G.bu = bu; G.bu2 = bu2; G.num = num; G.den = den;
        G.per = (n + (size_t)threads * 4 - 1) / ((size_t)threads * 4); if (G.per < 256) G.per = 256;
        fp_pool_run(pool, fp_gp_terms_task, &G, (int)((n + G.per - 1) / G.per));
        fr_t run = F->one;
        for (size_t i = 0; i < n; i++) { pre[i] = run; f4_mul(F, &run, &run, &den[i]); }
        fr_t inv; f4_inv(F, &inv, &run);
        for (size_t i = n; i-- > 0;) { fr_t di; f4_mul(F, &di, &inv, &pre[i]); f4_mul(F, &inv, &inv, &den[i]); f4_mul(F, &num[i], &num[i], &di); }
        wc[3][0] = F->one;
        for (size_t i = 0; i + 1 < n; i++) f4_mul(F, &wc[3][i + 1], &wc[3][i], &num[i]);
        free(num); free(den); free(pre);
    }
    fp_ifft_n(X, wc[3], pool, threads);
    for (int k = 0; k < 3; k++) { f4_sub(F, &wc[3][k], &wc[3][k], &bl[6 + k]); f4_add(F, &wc[3][n + k], &wc[3][n + k], &bl[6 + k]); }
    uint8_t z_pt[96], z_b[96];
    fp_commit(cv, X->srs, wc[3], n + 3, pool, threads, z_pt); g1_raw(cv, z_pt, z_b);
    {
        const uint8_t* parts[3]; size_t lens[3]; int np = 0;
        for (uint32_t k = 0; k < nbc; k++) { parts[np] = bsb_b[k]; lens[np++] = PT; }
        parts[np] = z_b; lens[np++] = PT;
        challenge("alpha", beta_raw, parts, lens, np, alpha_raw);
    }
    fr_t alpha; fr_from_be_reduce(F, &alpha, alpha_raw);

    /* ---- round 3: quotient on the coset (only l, r, o, Z are transformed: the trace sits there already) ---- */
    fr_t* ew[4];
    for (int j = 0; j < 4; j++) { ew[j] = fp_alloc(n4); fp_coset_eval(X, wc[j], j < 3 ? n + 2 : n + 3, ew[j], pool, threads); }
    fr_t* h = fp_alloc(n4);
    {
        fp_quot_job Q; Q.X = X; Q.el = ew[0]; Q.er = ew[1]; Q.eo = ew[2]; Q.ez = ew[3]; Q.h = h;
        Q.alpha = alpha; f4_sqr(F, &Q.a2, &alpha); Q.beta = beta; Q.gamma = gamma; Q.bu = bu; Q.bu2 = bu2;
        for (int j = 0; j < X->n_inj; j++) {
            const fr_t* written = (uint32_t)j < X->nb_public ? &pub[j] : &cval[j - X->nb_public];
            f4_sub(F, &Q.delta[j], written, &X->tl[FQK][X->inj_row[j]]);
        }
        Q.epi2[0] = epi2[0]; Q.epi2[1] = epi2[1];
        Q.per = (n4 + (size_t)threads * 4 - 1) / ((size_t)threads * 4); if (Q.per < 256) Q.per = 256;
        fp_pool_run(pool, fp_quot_task, &Q, (int)((n4 + Q.per - 1) / Q.per));
        fp_fft(cv, h, n4, X->w1i, pool, threads);
        fp_scale(cv, h, n4, X->uinv_pow, NULL, pool, threads);
    }
    for (int j = 0; j < 4; j++) free(ew[j]);
    for (uint32_t k = 0; k < nbc; k++) free(epi2[k]);
    int rc = 0;
    for (size_t i = 3 * (n + 2); i < n4; i++) if (!f4_is_zero(&h[i])) { rc = 4; break; }
    uint8_t h_pt[3][96], h_b[3][96];
    for (int j = 0; j < 3; j++) { fp_commit(cv, X->srs, h + (size_t)j * (n + 2), n + 2, pool, threads, h_pt[j]); g1_raw(cv, h_pt[j], h_b[j]); }
    { const uint8_t* parts[3] = {h_b[0], h_b[1], h_b[2]}; size_t lens[3] = {PT, PT, PT}; challenge("zeta", alpha_raw, parts, lens, 3, zeta_raw); }
    fr_t zeta; fr_from_be_reduce(F, &zeta, zeta_raw);

    /* ---- round 4 ---- */
    fr_t zw; f4_mul(F, &zw, &zeta, &X->omega);
    fr_t zshift, lz, rz, oz, s1z, s2z;
    fp_poly_eval(cv, &zshift, wc[3], n + 3, &zw, pool, threads);
    fp_poly_eval(cv, &lz, wc[0], n + 2, &zeta, pool, threads); fp_poly_eval(cv, &rz, wc[1], n + 2, &zeta, pool, threads);
    fp_poly_eval(cv, &oz, wc[2], n + 2, &zeta, pool, threads);
    fp_poly_eval(cv, &s1z, X->tc[FS1], n, &zeta, pool, threads); fp_poly_eval(cv, &s2z, X->tc[FS2], n, &zeta, pool, threads);
    fr_t qcpz[2];
    for (uint32_t k = 0; k < nbc; k++) fp_poly_eval(cv, &qcpz[k], X->qcp_c[k], n, &zeta, pool, threads);
    fr_t* q2 = fp_alloc(n + 3);
    poly_div_linear(F, q2, wc[3], n + 3, &zw);
    fr_t a2, zn, lag0, c_s3, c_z, zn2, zn2sq, t, a, b, c;
    f4_sqr(F, &a2, &alpha);
    f4_pow_u64(F, &zn, &zeta, n); f4_sub(F, &zn, &zn, &F->one);
    f4_sub(F, &t, &zeta, &F->one); f4_inv(F, &t, &t); f4_mul(F, &lag0, &zn, &X->ninv); f4_mul(F, &lag0, &lag0, &t);
    f4_mul(F, &t, &beta, &s1z); f4_add(F, &a, &lz, &t); f4_add(F, &a, &a, &gamma);
    f4_mul(F, &t, &beta, &s2z); f4_add(F, &b, &rz, &t); f4_add(F, &b, &b, &gamma);
    f4_mul(F, &c_s3, &alpha, &beta); f4_mul(F, &c_s3, &c_s3, &zshift); f4_mul(F, &c_s3, &c_s3, &a); f4_mul(F, &c_s3, &c_s3, &b);
    /* lin(zeta) from the verifier's identity (templateLogicSigBN254.go:203-218): -(PI(zeta) + alpha z(wz) (..)(..)(o + gamma) - alpha^2 L_0) */
    fr_t linz;
    {
        fr_t piz; memset(&piz, 0, sizeof piz);
        for (int j = 0; j < X->n_inj; j++) {   /* L_r(zeta) = omega^r (zeta^n - 1) / (n (zeta - omega^r)) over the rows a proof writes */
            const uint32_t r = X->inj_row[j];
            const fr_t* written = (uint32_t)j < X->nb_public ? &pub[j] : &cval[j - X->nb_public];
            fr_t d, lr; f4_sub(F, &d, &zeta, &X->omega_pow[r]); f4_inv(F, &d, &d);
            f4_mul(F, &lr, &zn, &X->ninv); f4_mul(F, &lr, &lr, &X->omega_pow[r]); f4_mul(F, &lr, &lr, &d);
            f4_mul(F, &lr, &lr, written); f4_add(F, &piz, &piz, &lr);
        }
        fr_t og, prod; f4_add(F, &og, &oz, &gamma);
        f4_mul(F, &prod, &alpha, &zshift); f4_mul(F, &prod, &prod, &a); f4_mul(F, &prod, &prod, &b); f4_mul(F, &prod, &prod, &og);
        f4_add(F, &linz, &piz, &prod);
        f4_mul(F, &t, &a2, &lag0); f4_sub(F, &linz, &linz, &t);
        f4_neg(F, &linz, &linz);
    }
    f4_mul(F, &t, &beta, &zeta); f4_add(F, &a, &lz, &t); f4_add(F, &a, &a, &gamma);
    f4_mul(F, &t, &t, &u); f4_add(F, &b, &rz, &t); f4_add(F, &b, &b, &gamma);
    f4_mul(F, &t, &t, &u); f4_add(F, &c, &oz, &t); f4_add(F, &c, &c, &gamma);
    f4_mul(F, &t, &a, &b); f4_mul(F, &t, &t, &c); f4_mul(F, &t, &t, &alpha);
    f4_mul(F, &c_z, &a2, &lag0); f4_sub(F, &c_z, &c_z, &t);
    f4_pow_u64(F, &zn2, &zeta, n + 2); f4_sqr(F, &zn2sq, &zn2);
    /* [lin] = sum coef_i [poly_i] over commitments in hand */
    fr_t lr, mz, mz2, mz3, one = F->one;
    f4_mul(F, &lr, &lz, &rz); f4_neg(F, &mz, &zn); f4_mul(F, &mz2, &mz, &zn2); f4_mul(F, &mz3, &mz, &zn2sq);
    uint8_t lin_pt[96], lin_b[96];
    {
        uint8_t pts[12][96];
        memcpy(pts[0], X->vk_pt[FQL], 96); memcpy(pts[1], X->vk_pt[FQR], 96); memcpy(pts[2], X->vk_pt[FQM], 96); memcpy(pts[3], X->vk_pt[FQO], 96);
        memcpy(pts[4], X->vk_pt[FQK], 96); memcpy(pts[5], X->vk_pt[FS3], 96); memcpy(pts[6], z_pt, 96); memcpy(pts[7], h_pt[0], 96);
        memcpy(pts[8], h_pt[1], 96); memcpy(pts[9], h_pt[2], 96);
        fr_t ks[12] = {lz, rz, lr, oz, one, c_s3, c_z, mz, mz2, mz3};
        for (uint32_t k = 0; k < nbc; k++) { memcpy(pts[10 + k], bsb_pt[k], 96); ks[10 + k] = qcpz[k]; }
        fp_small_msm(cv, pts, ks, 10 + (int)nbc, lin_pt);
        g1_raw(cv, lin_pt, lin_b);
    }
    uint8_t gk_raw[32];
    fr_t claimed[8] = {linz, lz, rz, oz, s1z, s2z};
    for (uint32_t k = 0; k < nbc; k++) claimed[6 + k] = qcpz[k];
    {
        uint8_t zeta_be[32], cvb[8][32], zsb[32];
        fr_to_be(F, &zeta, zeta_be);
        for (uint32_t i = 0; i < 6 + nbc; i++) fr_to_be(F, &claimed[i], cvb[i]);
        fr_to_be(F, &zshift, zsb);
        const uint8_t* parts[20]; size_t lens[20]; int np = 0;
        parts[np] = zeta_be; lens[np++] = 32;
        parts[np] = lin_b; lens[np++] = PT;
        for (int j = 0; j < 3; j++) { parts[np] = lro_b[j]; lens[np++] = PT; }
        parts[np] = X->vkb[FS1]; lens[np++] = PT; parts[np] = X->vkb[FS2]; lens[np++] = PT;
        for (uint32_t k = 0; k < nbc; k++) { parts[np] = X->qcp_b[k]; lens[np++] = PT; }
        for (uint32_t i = 0; i < 6 + nbc; i++) { parts[np] = cvb[i]; lens[np++] = 32; }
        parts[np] = zsb; lens[np++] = 32;
        challenge("gamma", NULL, parts, lens, np, gk_raw);
    }
    fr_t gk; fr_from_be_reduce(F, &gk, gk_raw);
    /* folded = lin + gk l + gk^2 r + gk^3 o + gk^4 S1 + gk^5 S2 in one pass over lin's constituents */
    fr_t* folded = fp_alloc(n + 3);
    {
        fr_t g1 = gk, g2, g3, g4, g5;
        f4_mul(F, &g2, &g1, &gk); f4_mul(F, &g3, &g2, &gk); f4_mul(F, &g4, &g3, &gk); f4_mul(F, &g5, &g4, &gk);
        /* (lin takes the TRACE's Qk, the polynomial behind the verifying key's [Qk]; the public inputs enter through PI(zeta)) */
        const fr_t* ps[19] = {X->tc[FQL], X->tc[FQR], X->tc[FQM], X->tc[FQO], X->tc[FQK], X->tc[FS3], wc[3], h, h + (n + 2), h + 2 * (n + 2),
                              wc[0], wc[1], wc[2], X->tc[FS1], X->tc[FS2]};
        size_t ls[19] = {n, n, n, n, n, n, n + 3, n + 2, n + 2, n + 2, n + 2, n + 2, n + 2, n, n};
        fr_t ks[19] = {lz, rz, lr, oz, one, c_s3, c_z, mz, mz2, mz3, g1, g2, g3, g4, g5};
        int cnt = 15;
        fr_t gpow = g5;
        for (uint32_t k = 0; k < nbc; k++) {        /* lin takes qcp_k(zeta) pi2_k; the fold takes gk^(6+k) Qcp_k */
            ps[cnt] = pi2c[k]; ls[cnt] = n; ks[cnt++] = qcpz[k];
            f4_mul(F, &gpow, &gpow, &gk);
            ps[cnt] = X->qcp_c[k]; ls[cnt] = n; ks[cnt++] = gpow;
        }
        fp_lc_job J; J.F = F; J.out = folded; J.ps = ps; J.ls = ls; J.ks = ks; J.count = cnt; J.n = n + 3;
        J.per = (n + 3 + (size_t)threads * 4 - 1) / ((size_t)threads * 4); if (J.per < 256) J.per = 256;
        fp_pool_run(pool, fp_lc_task, &J, (int)((n + 3 + J.per - 1) / J.per));
    }
    fr_t* q1 = fp_alloc(n + 3);
    poly_div_linear(F, q1, folded, n + 3, &zeta);
    uint8_t bh_pt[96], bh_b[96], zs_pt[96], zs_b[96];
    fp_commit(cv, X->srs, q1, n + 2, pool, threads, bh_pt); g1_raw(cv, bh_pt, bh_b);
    fp_commit(cv, X->srs, q2, n + 2, pool, threads, zs_pt); g1_raw(cv, zs_pt, zs_b);

    uint8_t* wp = blob;
    for (int j = 0; j < 3; j++) { memcpy(wp, lro_b[j], PT); wp += PT; }
    for (int j = 0; j < 3; j++) { memcpy(wp, h_b[j], PT); wp += PT; }
    for (int i = 1; i < 6; i++) { fr_to_be(F, &claimed[i], wp); wp += 32; }
    memcpy(wp, z_b, PT); wp += PT;
    fr_to_be(F, &zshift, wp); wp += 32;
    memcpy(wp, bh_b, PT); wp += PT;
    memcpy(wp, zs_b, PT); wp += PT;
    for (uint32_t k = 0; k < nbc; k++) { fr_to_be(F, &claimed[6 + k], wp); wp += 32; }      /* helper.go:74-85 */
    for (uint32_t k = 0; k < nbc; k++) { memcpy(wp, bsb_b[k], PT); wp += PT; }
    *blob_len = (uint64_t)(wp - blob);
    if (challenges_out) {
        fr_to_be(F, &gamma, challenges_out); fr_to_be(F, &beta, challenges_out + 32); fr_to_be(F, &alpha, challenges_out + 64);
        fr_to_be(F, &zeta, challenges_out + 96); fr_to_be(F, &gk, challenges_out + 128);
    }
    for (int j = 0; j < 4; j++) free(wc[j]);
    for (uint32_t k = 0; k < nbc; k++) free(pi2c[k]);
    free(pub_b); free(h); free(q1); free(q2); free(folded);
    fp_pool_destroy(pool);
    return rc;
}
