/* ORACLE - TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the PLONK prover path AlgoPlonk reaches through plonk.Prove
 * (/root/reference/algoplonk.go:89; second call site /root/reference/testutils/testutils.go:47).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; libapk never links it.
 *
 * The arithmetic lives in gnark v0.15.0 / gnark-crypto v0.20.1 (/root/reference/go.mod:8-9), which are NOT vendored
 * and not on this machine, so this restates the published algorithms (Montgomery CIOS, Jacobian G1, Pippenger with
 * per-window buckets, radix-2 Cooley-Tukey NTT, KZG) and the round structure of SURVEY.md §3.3; everything the
 * verifier can see is pinned by /root/reference/verifier/templateLogicSigBN254.go (lines cited below).
 *
 * PARITY STATUS: "parity unpinned" against gnark at the value level (the reference holds no golden
 * proof bytes).  This file is pinned against oracle/plonk.py (independent textbook route on plain Python ints) by
 * tests/test_oracle_c.py, and oracle/plonk.py is pinned by the verifier transcribed from the reference's template.
 *
 * Layouts = gnark in-memory (little-endian limbs, Montgomery form; affine X||Y, (0,0) = infinity), identical to
 * include/apk.h so the same buffers feed both sides.
 */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ---- tiny parallel-for ------------------------------------------------------------------------------------ */
typedef void (*task_fn)(void* arg, int index);
typedef struct { task_fn fn; void* arg; int next, count; pthread_mutex_t mu; } pf_state;
static void* pf_worker(void* p) {
    pf_state* s = (pf_state*)p;
    for (;;) {
        pthread_mutex_lock(&s->mu);
        int i = s->next++;
        pthread_mutex_unlock(&s->mu);
        if (i >= s->count) return NULL;
        s->fn(s->arg, i);
    }
}
static void parallel_for(task_fn fn, void* arg, int count, int threads) {
    if (threads <= 1 || count <= 1) { for (int i = 0; i < count; i++) fn(arg, i); return; }
    if (threads > count) threads = count;
    if (threads > 256) threads = 256;
    pf_state s = {fn, arg, 0, count, PTHREAD_MUTEX_INITIALIZER};
    pthread_t th[256];
    for (int t = 0; t < threads; t++) pthread_create(&th[t], NULL, pf_worker, &s);
    for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
}

/* ---- field / curve instantiations ------------------------------------------------------------------------- */
#define NL 4
#define FN(x) f4_##x
#include "field_tmpl.h"
#undef NL
#undef FN
#define NL 6
#define FN(x) f6_##x
#include "field_tmpl.h"
#undef NL
#undef FN

#define FPN(x) f4_##x
#define CN(x) bn_##x
#include "curve_tmpl.h"
#undef FPN
#undef CN
#define FPN(x) f6_##x
#define CN(x) bls_##x
#include "curve_tmpl.h"
#undef FPN
#undef CN

typedef f4_t fr_t;
typedef f4_field fr_field;

/* moduli: /root/reference/verifier/templateLogicSigBN254.go:15,18 and templateLogicSigBLS12_381.go:15,18 */
static const uint64_t R_BN[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
static const uint64_t P_BN[4] = {0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
static const uint64_t R_BLS[4] = {0xffffffff00000001ull, 0x53bda402fffe5bfeull, 0x3339d80809a1d805ull, 0x73eda753299d7d48ull};
static const uint64_t P_BLS[6] = {0xb9feffffffffaaabull, 0x1eabfffeb153ffffull, 0x6730d2a0f6b0f624ull, 0x64774b84f38512bfull, 0x4b1ba7b6434bacd7ull, 0x1a0111ea397fe69aull};
/* 2-adic roots of unity [UPSTREAM gnark-crypto fr/fft], canonical limbs; orders 2^28 / 2^32 checked in init */
static const uint64_t ROOT_BN[4] = {0x9bd61b6e725b19f0ull, 0x402d111e41112ed4ull, 0x00e0a7eb8ef62abcull, 0x2a3c09f0a58a7e85ull};
static const uint64_t ROOT_BLS[4] = {0x3829971f439f0d2bull, 0xb63683508c2280b9ull, 0xd09b681922c813b4ull, 0x16a2a19edfe81f20ull};

static fr_field FR[2];
static f4_field FP_BN;
static f6_field FP_BLS;
static fr_t ROOT[2];
static int ADICITY[2] = {28, 32};
static uint64_t SHIFT[2] = {5, 7};
static int SCALAR_BITS[2] = {254, 255};
static int g_init = 0;

static uint64_t neg_inv64(uint64_t p0) {
    uint64_t x = 1;
    for (int i = 0; i < 6; i++) x *= 2 - p0 * x; /* Newton: x = p0^-1 mod 2^64 */
    return (uint64_t)0 - x;
}
static void f4_field_init(f4_field* F, const uint64_t* mod) {
    memcpy(F->mod.l, mod, 32);
    F->inv = neg_inv64(mod[0]);
    f4_t one; memset(&one, 0, sizeof one); one.l[0] = 1;
    f4_t x = one;
    for (int i = 0; i < 256; i++) { uint64_t c = f4_add_raw(&x, &x, &x); if (c || f4_geq(&x, &F->mod)) f4_sub_raw(&x, &x, &F->mod); }
    F->one = x;
    for (int i = 0; i < 256; i++) { uint64_t c = f4_add_raw(&x, &x, &x); if (c || f4_geq(&x, &F->mod)) f4_sub_raw(&x, &x, &F->mod); }
    F->r2 = x;
}
static void f6_field_init(f6_field* F, const uint64_t* mod) {
    memcpy(F->mod.l, mod, 48);
    F->inv = neg_inv64(mod[0]);
    f6_t one; memset(&one, 0, sizeof one); one.l[0] = 1;
    f6_t x = one;
    for (int i = 0; i < 384; i++) { uint64_t c = f6_add_raw(&x, &x, &x); if (c || f6_geq(&x, &F->mod)) f6_sub_raw(&x, &x, &F->mod); }
    F->one = x;
    for (int i = 0; i < 384; i++) { uint64_t c = f6_add_raw(&x, &x, &x); if (c || f6_geq(&x, &F->mod)) f6_sub_raw(&x, &x, &F->mod); }
    F->r2 = x;
}
static void orc_init(void) {
    if (g_init) return;
    f4_field_init(&FR[0], R_BN);
    f4_field_init(&FR[1], R_BLS);
    f4_field_init(&FP_BN, P_BN);
    f6_field_init(&FP_BLS, P_BLS);
    fr_t t;
    memcpy(t.l, ROOT_BN, 32); f4_to_mont(&FR[0], &ROOT[0], &t);
    memcpy(t.l, ROOT_BLS, 32); f4_to_mont(&FR[1], &ROOT[1], &t);
    g_init = 1;
}

static void fr_set_u64(const fr_field* F, fr_t* r, uint64_t v) {
    fr_t t; memset(&t, 0, sizeof t); t.l[0] = v;
    f4_to_mont(F, r, &t);
}

/* ---- SHA-256 (FIPS 180-4) for the transcript --------------------------------------------------------------- */
typedef struct { uint32_t h[8]; uint64_t len; uint8_t buf[64]; size_t fill; } sha_t;
static const uint32_t SHA_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
    0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
    0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
    0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
    0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
    0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
#define ROTR(x, n) (((x) >> (n)) | ((x) << (32 - (n))))
static void sha_block(sha_t* s, const uint8_t* p) {
    uint32_t w[64];
    for (int i = 0; i < 16; i++) w[i] = (uint32_t)p[4 * i] << 24 | (uint32_t)p[4 * i + 1] << 16 | (uint32_t)p[4 * i + 2] << 8 | p[4 * i + 3];
    for (int i = 16; i < 64; i++) {
        uint32_t s0 = ROTR(w[i - 15], 7) ^ ROTR(w[i - 15], 18) ^ (w[i - 15] >> 3);
        uint32_t s1 = ROTR(w[i - 2], 17) ^ ROTR(w[i - 2], 19) ^ (w[i - 2] >> 10);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = s->h[0], b = s->h[1], c = s->h[2], d = s->h[3], e = s->h[4], f = s->h[5], g = s->h[6], h = s->h[7];
    for (int i = 0; i < 64; i++) {
        uint32_t t1 = h + (ROTR(e, 6) ^ ROTR(e, 11) ^ ROTR(e, 25)) + ((e & f) ^ (~e & g)) + SHA_K[i] + w[i];
        uint32_t t2 = (ROTR(a, 2) ^ ROTR(a, 13) ^ ROTR(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
        h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    s->h[0] += a; s->h[1] += b; s->h[2] += c; s->h[3] += d; s->h[4] += e; s->h[5] += f; s->h[6] += g; s->h[7] += h;
}
static void sha_init(sha_t* s) {
    static const uint32_t iv[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    memcpy(s->h, iv, sizeof iv); s->len = 0; s->fill = 0;
}
static void sha_update(sha_t* s, const void* data, size_t n) {
    const uint8_t* p = (const uint8_t*)data;
    s->len += n;
    while (n) {
        size_t take = 64 - s->fill; if (take > n) take = n;
        memcpy(s->buf + s->fill, p, take); s->fill += take; p += take; n -= take;
        if (s->fill == 64) { sha_block(s, s->buf); s->fill = 0; }
    }
}
static void sha_final(sha_t* s, uint8_t out[32]) {
    uint64_t bits = s->len * 8; uint8_t pad = 0x80, z = 0, lb[8];
    sha_update(s, &pad, 1);
    while (s->fill != 56) sha_update(s, &z, 1);
    for (int i = 0; i < 8; i++) lb[i] = (uint8_t)(bits >> (56 - 8 * i));
    sha_update(s, lb, 8);
    for (int i = 0; i < 8; i++) { out[4 * i] = s->h[i] >> 24; out[4 * i + 1] = s->h[i] >> 16; out[4 * i + 2] = s->h[i] >> 8; out[4 * i + 3] = s->h[i]; }
}

/* ---- byte codecs ------------------------------------------------------------------------------------------ */
static void fr_to_be(const fr_field* F, const fr_t* m, uint8_t* be) {
    fr_t c; f4_from_mont(F, &c, m);
    for (int i = 0; i < 4; i++) for (int b = 0; b < 8; b++) be[31 - (8 * i + b)] = (uint8_t)(c.l[i] >> (8 * b));
}
/* 32 big-endian bytes (any 256-bit value) reduced mod r, Montgomery form */
static void fr_from_be_reduce(const fr_field* F, fr_t* out, const uint8_t* be) {
    fr_t a;
    for (int i = 0; i < 4; i++) { a.l[i] = 0; for (int b = 0; b < 8; b++) a.l[i] |= (uint64_t)be[31 - (8 * i + b)] << (8 * b); }
    while (f4_geq(&a, &F->mod)) f4_sub_raw(&a, &a, &F->mod);
    f4_to_mont(F, out, &a);
}
/* gnark RawBytes(): X||Y big-endian; infinity = 0x40 then zeros on BLS12-381 (verifier/verifier.go:95-99), all zeros on BN254
 * (templateLogicSigBN254.go:57-61,131-132: one constant for the transcript and the ec ops; pinned by tests/golden/template_verdicts.json) */
static void g1_raw(int curve, const void* aff, uint8_t* out) {
    if (curve == 0) {
        const bn_aff* p = (const bn_aff*)aff;
        if (bn_aff_is_inf(p)) { memset(out, 0, 64); return; }
        f4_t c;
        f4_from_mont(&FP_BN, &c, &p->x); for (int i = 0; i < 4; i++) for (int b = 0; b < 8; b++) out[31 - (8 * i + b)] = (uint8_t)(c.l[i] >> (8 * b));
        f4_from_mont(&FP_BN, &c, &p->y); for (int i = 0; i < 4; i++) for (int b = 0; b < 8; b++) out[63 - (8 * i + b)] = (uint8_t)(c.l[i] >> (8 * b));
    } else {
        const bls_aff* p = (const bls_aff*)aff;
        if (bls_aff_is_inf(p)) { memset(out, 0, 96); out[0] = 0x40; return; }
        f6_t c;
        f6_from_mont(&FP_BLS, &c, &p->x); for (int i = 0; i < 6; i++) for (int b = 0; b < 8; b++) out[47 - (8 * i + b)] = (uint8_t)(c.l[i] >> (8 * b));
        f6_from_mont(&FP_BLS, &c, &p->y); for (int i = 0; i < 6; i++) for (int b = 0; b < 8; b++) out[95 - (8 * i + b)] = (uint8_t)(c.l[i] >> (8 * b));
    }
}
static size_t g1_size(int curve) { return curve == 0 ? 64 : 96; }

/* ---- MSM entry (kzg.Commit) -------------------------------------------------------------------------------- */
static void commit(int curve, const void* srs, const fr_t* coeffs, size_t n, int threads, void* out_aff) {
    const fr_field* F = &FR[curve];
    uint64_t* plain = (uint64_t*)malloc(n * 32);
    for (size_t i = 0; i < n; i++) { fr_t c; f4_from_mont(F, &c, &coeffs[i]); memcpy(plain + 4 * i, c.l, 32); }
    if (curve == 0) bn_msm(&FP_BN, (const bn_aff*)srs, plain, n, SCALAR_BITS[0], threads, (bn_aff*)out_aff);
    else bls_msm(&FP_BLS, (const bls_aff*)srs, plain, n, SCALAR_BITS[1], threads, (bls_aff*)out_aff);
    free(plain);
}

/* ---- NTT: natural in / natural out, iterative radix-2 DIT after a bit-reversal permutation ------------------ */
static void ntt_inplace(const fr_field* F, fr_t* a, size_t n, const fr_t* w /* omega^i, i < n/2 */) {
    int lg = 0; while (((size_t)1 << lg) < n) lg++;
    for (size_t i = 0; i < n; i++) {
        size_t j = 0;
        for (int b = 0; b < lg; b++) j |= ((i >> b) & 1) << (lg - 1 - b);
        if (i < j) { fr_t t = a[i]; a[i] = a[j]; a[j] = t; }
    }
    for (size_t len = 2; len <= n; len <<= 1) {
        size_t half = len >> 1, step = n / len;
        for (size_t s = 0; s < n; s += len)
            for (size_t j = 0; j < half; j++) {
                fr_t u = a[s + j], v;
                f4_mul(F, &v, &a[s + j + half], &w[j * step]);
                f4_add(F, &a[s + j], &u, &v);
                f4_sub(F, &a[s + j + half], &u, &v);
            }
    }
}
typedef struct { fr_t *w, *wi; fr_t ninv; size_t n; } domain_t;
static void domain_init(int curve, domain_t* d, size_t n) {
    const fr_field* F = &FR[curve];
    int lg = 0; while (((size_t)1 << lg) < n) lg++;
    fr_t g = ROOT[curve];
    for (int i = 0; i < ADICITY[curve] - lg; i++) f4_sqr(F, &g, &g);
    fr_t gi; f4_inv(F, &gi, &g);
    d->n = n;
    d->w = (fr_t*)malloc((n / 2 + 1) * sizeof(fr_t)); d->wi = (fr_t*)malloc((n / 2 + 1) * sizeof(fr_t));
    d->w[0] = F->one; d->wi[0] = F->one;
    for (size_t i = 1; i < n / 2; i++) { f4_mul(F, &d->w[i], &d->w[i - 1], &g); f4_mul(F, &d->wi[i], &d->wi[i - 1], &gi); }
    fr_t nn; fr_set_u64(F, &nn, n); f4_inv(F, &d->ninv, &nn);
}
static void domain_free(domain_t* d) { free(d->w); free(d->wi); }
static void fft(int curve, const domain_t* d, fr_t* a) { ntt_inplace(&FR[curve], a, d->n, d->w); }
static void ifft(int curve, const domain_t* d, fr_t* a) {
    ntt_inplace(&FR[curve], a, d->n, d->wi);
    for (size_t i = 0; i < d->n; i++) f4_mul(&FR[curve], &a[i], &a[i], &d->ninv);
}
static fr_t domain_gen(int curve, const domain_t* d) { return d->n > 1 ? d->w[1] : FR[curve].one; }

/* ---- exported primitives ---------------------------------------------------------------------------------- */
int orc_msm(int curve, const void* points, const void* scalars, uint64_t n, int threads, void* out) {
    orc_init();
    if (curve != 0 && curve != 1) return 1;
    commit(curve, points, (const fr_t*)scalars, n, threads, out);
    return 0;
}
/* which-agnostic: transforms `n` elements in place; coset uses the curve's CosetShift (5 / 7) */
int orc_ntt(int curve, void* data, uint64_t n, int inverse, int coset) {
    orc_init();
    if (curve != 0 && curve != 1) return 1;
    const fr_field* F = &FR[curve];
    domain_t d; domain_init(curve, &d, n);
    fr_t* a = (fr_t*)data;
    fr_t u; fr_set_u64(F, &u, SHIFT[curve]);
    if (!inverse) {
        if (coset) { fr_t p = F->one; for (uint64_t i = 0; i < n; i++) { f4_mul(F, &a[i], &a[i], &p); f4_mul(F, &p, &p, &u); } }
        fft(curve, &d, a);
    } else {
        ifft(curve, &d, a);
        if (coset) { fr_t ui, p = F->one; f4_inv(F, &ui, &u); for (uint64_t i = 0; i < n; i++) { f4_mul(F, &a[i], &a[i], &p); f4_mul(F, &p, &p, &ui); } }
    }
    domain_free(&d);
    return 0;
}

/* ---- prover ------------------------------------------------------------------------------------------------ */
typedef struct {
    int curve; uint64_t n; uint32_t nb_public;
    const void *srs, *ql, *qr, *qm, *qo, *qk; const int64_t* perm;
} orc_circuit;

static void poly_eval(const fr_field* F, fr_t* r, const fr_t* c, size_t len, const fr_t* x) {
    fr_t acc; memset(&acc, 0, sizeof acc);
    for (size_t i = len; i-- > 0;) { f4_mul(F, &acc, &acc, x); f4_add(F, &acc, &acc, &c[i]); }
    *r = acc;
}
/* q = (f - f(z)) / (X - z), synthetic division; q has len-1 coefficients */
static void poly_div_linear(const fr_field* F, fr_t* q, const fr_t* f, size_t len, const fr_t* z) {
    fr_t acc; memset(&acc, 0, sizeof acc);
    for (size_t i = len - 1; i >= 1; i--) { f4_mul(F, &acc, &acc, z); f4_add(F, &acc, &acc, &f[i]); q[i - 1] = acc; }
}
static void challenge(const char* name, const uint8_t* prev, const uint8_t** parts, const size_t* lens, int np, uint8_t out[32]) {
    sha_t s; sha_init(&s);
    sha_update(&s, name, strlen(name));
    if (prev) sha_update(&s, prev, 32);
    for (int i = 0; i < np; i++) sha_update(&s, parts[i], lens[i]);
    sha_final(&s, out);
}

typedef struct { int curve; const domain_t* d; fr_t** polys; int inverse; } fft_job;
static void fft_task(void* arg, int i) { fft_job* J = (fft_job*)arg; if (J->inverse) ifft(J->curve, J->d, J->polys[i]); else fft(J->curve, J->d, J->polys[i]); }


typedef struct {
    int cv; size_t n4, chunk; fr_t** ev; fr_t* h; fr_t alpha, a2, beta, gamma, bu, bu2, u, w4; const fr_t* zhinv;
} quot_job;
static void quot_task(void* arg, int t) {
    quot_job* Q = (quot_job*)arg;
    const fr_field* F = &FR[Q->cv];
    enum { EL, ER, EO, EZ, EQK, EQL, EQR, EQM, EQO, ES1, ES2, ES3, EL0 };
    fr_t** ev = Q->ev; fr_t* h = Q->h;
    const size_t n4 = Q->n4, lo = (size_t)t * Q->chunk, hi = lo + Q->chunk < n4 ? lo + Q->chunk : n4;
    const fr_t alpha = Q->alpha, a2 = Q->a2, beta = Q->beta, gamma = Q->gamma, bu = Q->bu, bu2 = Q->bu2, w4 = Q->w4;
    const fr_t* zhinv = Q->zhinv;
    fr_t x; f4_pow_u64(F, &x, &w4, lo); f4_mul(F, &x, &x, &Q->u);
    for (size_t i = lo; i < hi; i++) {
        fr_t l = ev[EL][i], r = ev[ER][i], o = ev[EO][i], z = ev[EZ][i], zs = ev[EZ][(i + 4) % n4];
        fr_t gate, t, lg, rg, og, pa, pb, a, b, c, loc, num;
        f4_mul(F, &gate, &ev[EQL][i], &l);
        f4_mul(F, &t, &ev[EQR][i], &r); f4_add(F, &gate, &gate, &t);
        f4_mul(F, &t, &l, &r); f4_mul(F, &t, &t, &ev[EQM][i]); f4_add(F, &gate, &gate, &t);
        f4_mul(F, &t, &ev[EQO][i], &o); f4_add(F, &gate, &gate, &t);
        f4_add(F, &gate, &gate, &ev[EQK][i]);
        f4_add(F, &lg, &l, &gamma); f4_add(F, &rg, &r, &gamma); f4_add(F, &og, &o, &gamma);
        f4_mul(F, &t, &beta, &ev[ES1][i]); f4_add(F, &a, &lg, &t);
        f4_mul(F, &t, &beta, &ev[ES2][i]); f4_add(F, &b, &rg, &t);
        f4_mul(F, &t, &beta, &ev[ES3][i]); f4_add(F, &c, &og, &t);
        f4_mul(F, &pa, &zs, &a); f4_mul(F, &pa, &pa, &b); f4_mul(F, &pa, &pa, &c);
        f4_mul(F, &t, &beta, &x); f4_add(F, &a, &lg, &t);
        f4_mul(F, &t, &bu, &x); f4_add(F, &b, &rg, &t);
        f4_mul(F, &t, &bu2, &x); f4_add(F, &c, &og, &t);
        f4_mul(F, &pb, &z, &a); f4_mul(F, &pb, &pb, &b); f4_mul(F, &pb, &pb, &c);
        f4_sub(F, &t, &z, &F->one); f4_mul(F, &loc, &ev[EL0][i], &t);
        f4_sub(F, &t, &pa, &pb); f4_mul(F, &t, &t, &alpha); f4_add(F, &num, &gate, &t);
        f4_mul(F, &t, &a2, &loc); f4_add(F, &num, &num, &t);
        f4_mul(F, &h[i], &num, &zhinv[i & 3]);
        f4_mul(F, &x, &x, &w4);
    }
}

/* plonk.Prove (algoplonk.go:89).  Inputs as in include/apk.h apk_prove; output = the proof blob of helper.go:13-88
 * (768 bytes BN254 / 1056 bytes BLS12-381, no BSB22 support in the C oracle) */
int orc_prove(const orc_circuit* C, const void* Lp, const void* Rp, const void* Op, const void* pubp, const void* blindp,
              int threads, uint8_t* blob, uint64_t* blob_len, uint8_t* challenges_out /* 5 x 32 BE or NULL */) {
    orc_init();
    const int cv = C->curve;
    if (cv != 0 && cv != 1) return 1;
    const fr_field* F = &FR[cv];
    const size_t n = C->n, n4 = 4 * n, PT = g1_size(cv);
    const fr_t *L = (const fr_t*)Lp, *R = (const fr_t*)Rp, *O = (const fr_t*)Op, *pub = (const fr_t*)pubp, *bl = (const fr_t*)blindp;
    domain_t d0, d1; domain_init(cv, &d0, n); domain_init(cv, &d1, n4);
    const fr_t w = domain_gen(cv, &d0);
    fr_t u, u2; fr_set_u64(F, &u, SHIFT[cv]); f4_sqr(F, &u2, &u);
    fr_t* omega_pow = (fr_t*)malloc(n * sizeof(fr_t));
    omega_pow[0] = F->one; for (size_t i = 1; i < n; i++) f4_mul(F, &omega_pow[i], &omega_pow[i - 1], &w);

    /* trace polynomials (plonk.Setup keeps them in the proving key; rebuilt here per call like gnark's NewTrace) */
    enum { QL, QR, QM, QO, QK, S1, S2, S3, NTRACE };
    fr_t* tl[NTRACE]; fr_t* tc[NTRACE];
    const void* cols[5] = {C->ql, C->qr, C->qm, C->qo, C->qk};
    for (int i = 0; i < NTRACE; i++) { tl[i] = (fr_t*)malloc(n * sizeof(fr_t)); tc[i] = (fr_t*)malloc(n * sizeof(fr_t)); }
    for (int i = 0; i < 5; i++) memcpy(tl[i], cols[i], n * sizeof(fr_t));
    for (int j = 0; j < 3; j++)
        for (size_t i = 0; i < n; i++) {
            int64_t p = C->perm[(size_t)j * n + i];
            size_t blk = (size_t)p / n, pos = (size_t)p % n;
            tl[S1 + j][i] = omega_pow[pos];
            if (blk == 1) f4_mul(F, &tl[S1 + j][i], &tl[S1 + j][i], &u);
            if (blk == 2) f4_mul(F, &tl[S1 + j][i], &tl[S1 + j][i], &u2);
        }
    for (int i = 0; i < NTRACE; i++) memcpy(tc[i], tl[i], n * sizeof(fr_t));
    { fft_job J = {cv, &d0, tc, 1}; parallel_for(fft_task, &J, NTRACE, threads); }
    /* VK commitments feed the transcript (templateLogicSigBN254.go:131-132) */
    uint8_t vkb[NTRACE][96];
    for (int i = 0; i < NTRACE; i++) { uint8_t pt[96]; commit(cv, C->srs, tc[i], n, threads, pt); g1_raw(cv, pt, vkb[i]); }

    /* ---- round 1: blinded wire polynomials (canonical, n+2 coefficients) and their commitments ----------- */
    fr_t* wc[4]; /* l, r, o, z blinded canonical, capacity n+3 */
    for (int j = 0; j < 4; j++) wc[j] = (fr_t*)calloc(n + 3, sizeof(fr_t));
    memcpy(wc[0], L, n * sizeof(fr_t)); memcpy(wc[1], R, n * sizeof(fr_t)); memcpy(wc[2], O, n * sizeof(fr_t));
    { fft_job J = {cv, &d0, wc, 1}; parallel_for(fft_task, &J, 3, threads); }
    for (int j = 0; j < 3; j++)
        for (int k = 0; k < 2; k++) { f4_sub(F, &wc[j][k], &wc[j][k], &bl[2 * j + k]); f4_add(F, &wc[j][n + k], &wc[j][n + k], &bl[2 * j + k]); }
    uint8_t lro_pt[3][96], lro_b[3][96];
    for (int j = 0; j < 3; j++) { commit(cv, C->srs, wc[j], n + 2, threads, lro_pt[j]); g1_raw(cv, lro_pt[j], lro_b[j]); }
    /* completed Qk */
    fr_t* qkf = (fr_t*)malloc(n * sizeof(fr_t));
    memcpy(qkf, tl[QK], n * sizeof(fr_t));
    for (uint32_t i = 0; i < C->nb_public; i++) qkf[i] = pub[i];
    ifft(cv, &d0, qkf);
    uint8_t* pub_b = (uint8_t*)malloc((size_t)C->nb_public * 32 + 1);
    for (uint32_t i = 0; i < C->nb_public; i++) fr_to_be(F, &pub[i], pub_b + 32 * i);

    uint8_t gamma_raw[32], beta_raw[32], alpha_raw[32], zeta_raw[32];
    {
        const uint8_t* parts[12] = {vkb[S1], vkb[S2], vkb[S3], vkb[QL], vkb[QR], vkb[QM], vkb[QO], vkb[QK], pub_b, lro_b[0], lro_b[1], lro_b[2]};
        size_t lens[12] = {PT, PT, PT, PT, PT, PT, PT, PT, (size_t)C->nb_public * 32, PT, PT, PT};
        challenge("gamma", NULL, parts, lens, 12, gamma_raw);
        challenge("beta", gamma_raw, NULL, NULL, 0, beta_raw);
    }
    fr_t gamma, beta; fr_from_be_reduce(F, &gamma, gamma_raw); fr_from_be_reduce(F, &beta, beta_raw);

    /* ---- round 2: grand product -------------------------------------------------------- */
    {
        fr_t *num = (fr_t*)malloc(n * sizeof(fr_t)), *den = (fr_t*)malloc(n * sizeof(fr_t)), *pre = (fr_t*)malloc(n * sizeof(fr_t));
        fr_t bu, bu2; f4_mul(F, &bu, &beta, &u); f4_mul(F, &bu2, &beta, &u2);
        fr_t run = F->one;
        for (size_t i = 0; i < n; i++) {
            fr_t l, r, o, t, a, b, c;
            f4_add(F, &l, &L[i], &gamma); f4_add(F, &r, &R[i], &gamma); f4_add(F, &o, &O[i], &gamma);
            f4_mul(F, &t, &beta, &omega_pow[i]); f4_add(F, &a, &l, &t);
            f4_mul(F, &t, &bu, &omega_pow[i]); f4_add(F, &b, &r, &t);
            f4_mul(F, &t, &bu2, &omega_pow[i]); f4_add(F, &c, &o, &t);
            f4_mul(F, &num[i], &a, &b); f4_mul(F, &num[i], &num[i], &c);
            f4_mul(F, &t, &beta, &tl[S1][i]); f4_add(F, &a, &l, &t);
            f4_mul(F, &t, &beta, &tl[S2][i]); f4_add(F, &b, &r, &t);
            f4_mul(F, &t, &beta, &tl[S3][i]); f4_add(F, &c, &o, &t);
            f4_mul(F, &den[i], &a, &b); f4_mul(F, &den[i], &den[i], &c);
            pre[i] = run; f4_mul(F, &run, &run, &den[i]);
        }
        fr_t inv; f4_inv(F, &inv, &run);
        for (size_t i = n; i-- > 0;) { fr_t di; f4_mul(F, &di, &inv, &pre[i]); f4_mul(F, &inv, &inv, &den[i]); f4_mul(F, &num[i], &num[i], &di); }
        wc[3][0] = F->one;
        for (size_t i = 0; i + 1 < n; i++) f4_mul(F, &wc[3][i + 1], &wc[3][i], &num[i]);
        free(num); free(den); free(pre);
    }
    ifft(cv, &d0, wc[3]);
    for (int k = 0; k < 3; k++) { f4_sub(F, &wc[3][k], &wc[3][k], &bl[6 + k]); f4_add(F, &wc[3][n + k], &wc[3][n + k], &bl[6 + k]); }
    uint8_t z_pt[96], z_b[96];
    commit(cv, C->srs, wc[3], n + 3, threads, z_pt); g1_raw(cv, z_pt, z_b);
    { const uint8_t* parts[1] = {z_b}; size_t lens[1] = {PT}; challenge("alpha", beta_raw, parts, lens, 1, alpha_raw); }
    fr_t alpha; fr_from_be_reduce(F, &alpha, alpha_raw);

    /* ---- round 3: quotient on the coset u*<omega_4n>  (identity: SURVEY.md App. E) ------------------------- */
    enum { EL, ER, EO, EZ, EQK, EQL, EQR, EQM, EQO, ES1, ES2, ES3, EL0, NEV };
    fr_t* ev[NEV];
    for (int i = 0; i < NEV; i++) ev[i] = (fr_t*)calloc(n4, sizeof(fr_t));
    memcpy(ev[EL], wc[0], (n + 2) * sizeof(fr_t)); memcpy(ev[ER], wc[1], (n + 2) * sizeof(fr_t)); memcpy(ev[EO], wc[2], (n + 2) * sizeof(fr_t));
    memcpy(ev[EZ], wc[3], (n + 3) * sizeof(fr_t)); memcpy(ev[EQK], qkf, n * sizeof(fr_t));
    memcpy(ev[EQL], tc[QL], n * sizeof(fr_t)); memcpy(ev[EQR], tc[QR], n * sizeof(fr_t)); memcpy(ev[EQM], tc[QM], n * sizeof(fr_t));
    memcpy(ev[EQO], tc[QO], n * sizeof(fr_t)); memcpy(ev[ES1], tc[S1], n * sizeof(fr_t)); memcpy(ev[ES2], tc[S2], n * sizeof(fr_t));
    memcpy(ev[ES3], tc[S3], n * sizeof(fr_t));
    for (size_t i = 0; i < n; i++) ev[EL0][i] = d0.ninv; /* L_0 = (1/n) sum X^i */
    {
        fr_t* upow = (fr_t*)malloc((n + 3) * sizeof(fr_t));
        upow[0] = F->one; for (size_t i = 1; i < n + 3; i++) f4_mul(F, &upow[i], &upow[i - 1], &u);
        for (int k = 0; k < NEV; k++) for (size_t i = 0; i < n + 3; i++) f4_mul(F, &ev[k][i], &ev[k][i], &upow[i]);
        free(upow);
        fft_job J = {cv, &d1, ev, 0}; parallel_for(fft_task, &J, NEV, threads);
    }
    fr_t* h = (fr_t*)malloc(n4 * sizeof(fr_t));
    {
        const fr_t w4 = domain_gen(cv, &d1);
        fr_t a2; f4_sqr(F, &a2, &alpha);
        fr_t bu, bu2; f4_mul(F, &bu, &beta, &u); f4_mul(F, &bu2, &beta, &u2);
        fr_t zhinv[4];
        { fr_t un, i4, cur; f4_pow_u64(F, &un, &u, n); f4_pow_u64(F, &i4, &w4, n); cur = un;
          for (int k = 0; k < 4; k++) { fr_t t; f4_sub(F, &t, &cur, &F->one); f4_inv(F, &zhinv[k], &t); f4_mul(F, &cur, &cur, &i4); } }
        {
            quot_job Q; Q.cv = cv; Q.n4 = n4; Q.ev = ev; Q.h = h; Q.alpha = alpha; Q.a2 = a2; Q.beta = beta; Q.gamma = gamma;
            Q.bu = bu; Q.bu2 = bu2; Q.u = u; Q.w4 = w4; Q.zhinv = zhinv;
            int tasks = threads > 1 ? threads * 2 : 1;
            Q.chunk = (n4 + tasks - 1) / tasks;
            parallel_for(quot_task, &Q, (int)((n4 + Q.chunk - 1) / Q.chunk), threads);
        }
        ifft(cv, &d1, h);
        fr_t ui, p = F->one; f4_inv(F, &ui, &u);
        for (size_t i = 0; i < n4; i++) { f4_mul(F, &h[i], &h[i], &p); f4_mul(F, &p, &p, &ui); }
    }
    for (int i = 0; i < NEV; i++) free(ev[i]);
    int rc = 0;
    for (size_t i = 3 * (n + 2); i < n4; i++) if (!f4_is_zero(&h[i])) rc = 4; /* witness does not satisfy the circuit */
    uint8_t h_pt[3][96], h_b[3][96];
    for (int j = 0; j < 3; j++) { commit(cv, C->srs, h + (size_t)j * (n + 2), n + 2, threads, h_pt[j]); g1_raw(cv, h_pt[j], h_b[j]); }
    { const uint8_t* parts[3] = {h_b[0], h_b[1], h_b[2]}; size_t lens[3] = {PT, PT, PT}; challenge("zeta", alpha_raw, parts, lens, 3, zeta_raw); }
    fr_t zeta; fr_from_be_reduce(F, &zeta, zeta_raw);

    /* ---- round 4: openings (templateLogicSigBN254.go:195-201,231-254,280-320) ------------------------------ */
    fr_t zw; f4_mul(F, &zw, &zeta, &w);
    fr_t zshift; poly_eval(F, &zshift, wc[3], n + 3, &zw);
    fr_t* q = (fr_t*)calloc(n + 3, sizeof(fr_t));
    poly_div_linear(F, q, wc[3], n + 3, &zw);
    uint8_t zs_pt[96], zs_b[96];
    commit(cv, C->srs, q, n + 2, threads, zs_pt); g1_raw(cv, zs_pt, zs_b);
    fr_t lz, rz, oz, s1z, s2z;
    poly_eval(F, &lz, wc[0], n + 2, &zeta); poly_eval(F, &rz, wc[1], n + 2, &zeta); poly_eval(F, &oz, wc[2], n + 2, &zeta);
    poly_eval(F, &s1z, tc[S1], n, &zeta); poly_eval(F, &s2z, tc[S2], n, &zeta);
    fr_t a2, zn, lag0, c_s3, c_z, zn2, zn2sq, t, a, b, c;
    f4_sqr(F, &a2, &alpha);
    f4_pow_u64(F, &zn, &zeta, n); f4_sub(F, &zn, &zn, &F->one);
    f4_sub(F, &t, &zeta, &F->one); f4_inv(F, &t, &t); f4_mul(F, &lag0, &zn, &d0.ninv); f4_mul(F, &lag0, &lag0, &t);
    f4_mul(F, &t, &beta, &s1z); f4_add(F, &a, &lz, &t); f4_add(F, &a, &a, &gamma);
    f4_mul(F, &t, &beta, &s2z); f4_add(F, &b, &rz, &t); f4_add(F, &b, &b, &gamma);
    f4_mul(F, &c_s3, &alpha, &beta); f4_mul(F, &c_s3, &c_s3, &zshift); f4_mul(F, &c_s3, &c_s3, &a); f4_mul(F, &c_s3, &c_s3, &b);
    f4_mul(F, &t, &beta, &zeta); f4_add(F, &a, &lz, &t); f4_add(F, &a, &a, &gamma);
    f4_mul(F, &t, &t, &u); f4_add(F, &b, &rz, &t); f4_add(F, &b, &b, &gamma);
    f4_mul(F, &t, &t, &u); f4_add(F, &c, &oz, &t); f4_add(F, &c, &c, &gamma);
    f4_mul(F, &t, &a, &b); f4_mul(F, &t, &t, &c); f4_mul(F, &t, &t, &alpha);
    f4_mul(F, &c_z, &a2, &lag0); f4_sub(F, &c_z, &c_z, &t);
    f4_pow_u64(F, &zn2, &zeta, n + 2); f4_sqr(F, &zn2sq, &zn2);
    fr_t* lin = (fr_t*)calloc(n + 3, sizeof(fr_t));
    {
        fr_t lr; f4_mul(F, &lr, &lz, &rz);
        for (size_t i = 0; i < n + 3; i++) {
            fr_t acc; memset(&acc, 0, sizeof acc);
            if (i < n) {
                f4_mul(F, &t, &lz, &tc[QL][i]); f4_add(F, &acc, &acc, &t);
                f4_mul(F, &t, &rz, &tc[QR][i]); f4_add(F, &acc, &acc, &t);
                f4_mul(F, &t, &lr, &tc[QM][i]); f4_add(F, &acc, &acc, &t);
                f4_mul(F, &t, &oz, &tc[QO][i]); f4_add(F, &acc, &acc, &t);
                f4_add(F, &acc, &acc, &tc[QK][i]);
                f4_mul(F, &t, &c_s3, &tc[S3][i]); f4_add(F, &acc, &acc, &t);
            }
            f4_mul(F, &t, &c_z, &wc[3][i]); f4_add(F, &acc, &acc, &t);
            if (i < n + 2) {
                fr_t hh = h[i], t2;
                f4_mul(F, &t2, &zn2, &h[(n + 2) + i]); f4_add(F, &hh, &hh, &t2);
                f4_mul(F, &t2, &zn2sq, &h[2 * (n + 2) + i]); f4_add(F, &hh, &hh, &t2);
                f4_mul(F, &hh, &hh, &zn); f4_sub(F, &acc, &acc, &hh);
            }
            lin[i] = acc;
        }
    }
    fr_t linz; poly_eval(F, &linz, lin, n + 3, &zeta);
    uint8_t lin_pt[96], lin_b[96];
    commit(cv, C->srs, lin, n + 3, threads, lin_pt); g1_raw(cv, lin_pt, lin_b);
    uint8_t gk_raw[32];
    fr_t claimed[6] = {linz, lz, rz, oz, s1z, s2z};
    {
        uint8_t zeta_be[32], cvb[6][32], zsb[32];
        fr_to_be(F, &zeta, zeta_be);
        for (int i = 0; i < 6; i++) fr_to_be(F, &claimed[i], cvb[i]);
        fr_to_be(F, &zshift, zsb);
        const uint8_t* parts[14] = {zeta_be, lin_b, lro_b[0], lro_b[1], lro_b[2], vkb[S1], vkb[S2], cvb[0], cvb[1], cvb[2], cvb[3], cvb[4], cvb[5], zsb};
        size_t lens[14] = {32, PT, PT, PT, PT, PT, PT, 32, 32, 32, 32, 32, 32, 32};
        challenge("gamma", NULL, parts, lens, 14, gk_raw);
    }
    fr_t gk; fr_from_be_reduce(F, &gk, gk_raw);
    fr_t* folded = (fr_t*)calloc(n + 3, sizeof(fr_t));
    {
        const fr_t* ps[6] = {lin, wc[0], wc[1], wc[2], tc[S1], tc[S2]};
        size_t ls[6] = {n + 3, n + 2, n + 2, n + 2, n, n};
        fr_t acc = F->one;
        for (int k = 0; k < 6; k++) {
            for (size_t i = 0; i < ls[k]; i++) { f4_mul(F, &t, &acc, &ps[k][i]); f4_add(F, &folded[i], &folded[i], &t); }
            f4_mul(F, &acc, &acc, &gk);
        }
    }
    poly_div_linear(F, q, folded, n + 3, &zeta);
    uint8_t bh_pt[96], bh_b[96];
    commit(cv, C->srs, q, n + 2, threads, bh_pt); g1_raw(cv, bh_pt, bh_b);

    /* ---- marshal: helper.go:13-24,27-88 -------------------------------------------------------------------- */
    uint8_t* wp = blob;
    for (int j = 0; j < 3; j++) { memcpy(wp, lro_b[j], PT); wp += PT; }
    for (int j = 0; j < 3; j++) { memcpy(wp, h_b[j], PT); wp += PT; }
    for (int i = 1; i < 6; i++) { fr_to_be(F, &claimed[i], wp); wp += 32; }
    memcpy(wp, z_b, PT); wp += PT;
    fr_to_be(F, &zshift, wp); wp += 32;
    memcpy(wp, bh_b, PT); wp += PT;
    memcpy(wp, zs_b, PT); wp += PT;
    *blob_len = (uint64_t)(wp - blob);
    if (challenges_out) {
        fr_to_be(F, &gamma, challenges_out); fr_to_be(F, &beta, challenges_out + 32); fr_to_be(F, &alpha, challenges_out + 64);
        fr_to_be(F, &zeta, challenges_out + 96); fr_to_be(F, &gk, challenges_out + 128);
    }
    for (int i = 0; i < NTRACE; i++) { free(tl[i]); free(tc[i]); }
    for (int j = 0; j < 4; j++) free(wc[j]);
    free(qkf); free(pub_b); free(h); free(q); free(lin); free(folded); free(omega_pow);
    domain_free(&d0); domain_free(&d1);
    return rc;
}

/* the performance-first twin of orc_prove / orc_msm (same bytes; bench.py's cpu_baseline): orc_fast_setup / orc_fast_prove /
 * orc_fast_free / orc_msm_fast */
#include "fast_prover.c"
