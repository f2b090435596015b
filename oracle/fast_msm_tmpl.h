/* ORACLE (test infrastructure, never linked into libapk): a PERFORMANCE-FIRST host MSM, instantiated per curve like
 * curve_tmpl.h (FPN = base-field prefix, CN = curve prefix).  The clarity-first CN(msm) of curve_tmpl.h stays the checker; this
 * one exists so that bench.py's cpu_baseline is a credible CPU stand-in and is itself held to CN(msm)'s
 * bytes by tests/test_oracle_c.py.
 *
 * What gnark-crypto v0.20.1 ecc/<curve>/multiexp.go does on a CPU [UPSTREAM, not vendored; SURVEY.md section 3.4], restated:
 *   - signed c-bit digits (half the buckets), c from a cost model over the task count;
 *   - tasks = windows x point chunks on a persistent thread pool;
 *   - BATCH-AFFINE bucket accumulation: additions into distinct buckets are collected and their slopes
 *     (y2 - y1) / (x2 - x1) share ONE field inversion (Montgomery's trick) - ~6 field products per addition where a Jacobian
 *     mixed addition takes 11; a point whose bucket is already in the batch waits in a small queue;
 *   - per task a running-sum reduction of the buckets (Jacobian), per window the chunks' sums, Horner over the windows.
 */

#define FM_BATCH 512
#define FM_QUEUE 2048

typedef struct {
    const FPN(field) * F;
    const CN(aff) * pts;
    const int32_t* dig;     /* [n][nwin] signed digits */
    size_t n;
    int c, nwin, nchunk;
    CN(jac) * part;         /* [nwin][nchunk] */
} CN(fm_job);

/* signed digits of every scalar, point-major [n][nwin], made once per MSM (in parallel): d_j in [-2^(c-1), 2^(c-1)] */
typedef struct { const uint64_t* sc; int32_t* dig; size_t n; int c, nwin; size_t per; } CN(fm_dig_job);
static void CN(fm_dig_task)(void* arg, int t) {
    CN(fm_dig_job)* D = (CN(fm_dig_job)*)arg;
    const size_t lo = (size_t)t * D->per, hi = lo + D->per < D->n ? lo + D->per : D->n;
    const int c = D->c;
    for (size_t i = lo; i < hi; i++) {
        const uint64_t* s = D->sc + 4 * i;
        int carry = 0;
        for (int j = 0; j < D->nwin; j++) {
            const int bit = j * c, word = bit >> 6, off = bit & 63;
            uint64_t v = word < 4 ? s[word] >> off : 0;
            if (off + c > 64 && word + 1 < 4) v |= s[word + 1] << (64 - off);
            v &= ((uint64_t)1 << c) - 1;
            int d = (int)v + carry;
            if (d > (1 << (c - 1))) { d -= 1 << c; carry = 1; } else carry = 0;
            D->dig[i * D->nwin + j] = d;
        }
    }
}

typedef struct {
    CN(aff) * B;                 /* buckets, (0,0) = empty */
    uint8_t* in_batch;           /* bucket has a pending addition in the current batch */
    uint32_t bk[FM_BATCH];       /* bucket of the pending addition */
    CN(aff) pt[FM_BATCH];        /* the point to add (sign applied) */
    int nb;
} CN(fm_acc);

/* run the pending additions: B[bk] += pt with one shared inversion */
static void CN(fm_flush)(const FPN(field) * F, CN(fm_acc) * A) {
    const int n = A->nb;
    if (!n) return;
    FPN(t) den[FM_BATCH], pre[FM_BATCH], num[FM_BATCH];
    uint8_t kind[FM_BATCH];      /* 0 generic, 1 doubling, 2 cancels to infinity */
    FPN(t) acc = F->one;
    for (int i = 0; i < n; i++) {
        CN(aff)* b = &A->B[A->bk[i]];
        const CN(aff)* p = &A->pt[i];
        kind[i] = 0;
        if (FPN(eq)(&b->x, &p->x)) {
            if (FPN(eq)(&b->y, &p->y) && !FPN(is_zero)(&p->y)) {        /* doubling: slope = 3 x^2 / 2 y */
                kind[i] = 1;
                FPN(t) xx; FPN(sqr)(F, &xx, &p->x);
                FPN(dbl)(F, &num[i], &xx); FPN(add)(F, &num[i], &num[i], &xx);
                FPN(dbl)(F, &den[i], &p->y);
            } else { kind[i] = 2; den[i] = F->one; }
        } else {
            FPN(sub)(F, &den[i], &p->x, &b->x);
            FPN(sub)(F, &num[i], &p->y, &b->y);
        }
        pre[i] = acc;
        FPN(mul)(F, &acc, &acc, &den[i]);
    }
    FPN(t) inv; FPN(inv)(F, &inv, &acc);
    for (int i = n - 1; i >= 0; i--) {
        FPN(t) dinv; FPN(mul)(F, &dinv, &inv, &pre[i]);
        FPN(mul)(F, &inv, &inv, &den[i]);
        CN(aff)* b = &A->B[A->bk[i]];
        A->in_batch[A->bk[i]] = 0;
        if (kind[i] == 2) { memset(b, 0, sizeof *b); continue; }
        const CN(aff)* p = &A->pt[i];
        FPN(t) lam, x3, y3, t;
        FPN(mul)(F, &lam, &num[i], &dinv);
        FPN(sqr)(F, &x3, &lam);
        FPN(sub)(F, &x3, &x3, &b->x); FPN(sub)(F, &x3, &x3, &p->x);
        FPN(sub)(F, &t, &b->x, &x3);
        FPN(mul)(F, &y3, &lam, &t);
        FPN(sub)(F, &y3, &y3, &b->y);
        b->x = x3; b->y = y3;
    }
    A->nb = 0;
}

/* B[k] += p (sign applied by the caller): empty bucket = copy; bucket busy in this batch = the caller queues the point */
static inline int CN(fm_push)(const FPN(field) * F, CN(fm_acc) * A, uint32_t k, const CN(aff) * p) {
    if (A->in_batch[k]) return 0;
    CN(aff)* b = &A->B[k];
    if (CN(aff_is_inf)(b)) { *b = *p; return 1; }
    A->bk[A->nb] = k; A->pt[A->nb] = *p; A->in_batch[k] = 1;
    if (++A->nb == FM_BATCH) CN(fm_flush)(F, A);
    return 1;
}

static void CN(fm_task)(void* arg, int t) {
    CN(fm_job)* J = (CN(fm_job)*)arg;
    const FPN(field)* F = J->F;
    const int c = J->c, w = t / J->nchunk, ch = t % J->nchunk;
    const size_t nbk = (size_t)1 << (c - 1);
    const size_t per = (J->n + J->nchunk - 1) / J->nchunk;
    const size_t lo = (size_t)ch * per < J->n ? (size_t)ch * per : J->n, hi = lo + per < J->n ? lo + per : J->n;
    /* the task's working set lives in the WORKER's scratch (fp_scratch: grown once per thread, reused by every task it runs):
     * a malloc / free of a few hundred KB per task is an mmap / munmap pair, and with 256 threads of one process in them the
     * kernel's address-space lock was where the time went (64 concurrent proofs took 41 s each before this) */
    const size_t need = sizeof(CN(fm_acc)) + nbk * sizeof(CN(aff)) + nbk + FM_QUEUE * (sizeof(uint32_t) + sizeof(CN(aff))) + 256;
    uint8_t* mem = (uint8_t*)fp_scratch(need);
    CN(fm_acc)* A = (CN(fm_acc)*)mem; mem += (sizeof(CN(fm_acc)) + 63) & ~(size_t)63;
    A->B = (CN(aff)*)mem; mem += nbk * sizeof(CN(aff));
    uint32_t* qk = (uint32_t*)mem; mem += FM_QUEUE * sizeof(uint32_t);
    CN(aff)* qp = (CN(aff)*)mem; mem += FM_QUEUE * sizeof(CN(aff));
    A->in_batch = mem;
    memset(A->B, 0, nbk * sizeof(CN(aff)));
    memset(A->in_batch, 0, nbk);
    A->nb = 0;
    int nq = 0;
    for (size_t i = lo; i < hi; i++) {
        if (CN(aff_is_inf)(&J->pts[i])) continue;
        const int d = J->dig[i * J->nwin + w];
        if (!d) continue;
        CN(aff) p = J->pts[i];
        if (d < 0) FPN(neg)(F, &p.y, &p.y);
        const uint32_t k = (uint32_t)((d < 0 ? -d : d) - 1);
        if (!CN(fm_push)(F, A, k, &p)) {
            qk[nq] = k; qp[nq] = p;
            if (++nq == FM_QUEUE) {              /* drain: flush, retry; what still collides stays queued */
                CN(fm_flush)(F, A);
                int keep = 0;
                for (int q = 0; q < nq; q++)
                    if (!CN(fm_push)(F, A, qk[q], &qp[q])) { qk[keep] = qk[q]; qp[keep] = qp[q]; keep++; }
                nq = keep;
                if (nq == FM_QUEUE) {            /* everything hits one bucket (all-equal scalars): one at a time */
                    CN(fm_flush)(F, A);
                    for (int q = 0; q < nq; q++) { while (!CN(fm_push)(F, A, qk[q], &qp[q])) CN(fm_flush)(F, A); }
                    nq = 0;
                }
            }
        }
    }
    while (nq) {
        CN(fm_flush)(F, A);
        int keep = 0;
        for (int q = 0; q < nq; q++)
            if (!CN(fm_push)(F, A, qk[q], &qp[q])) { qk[keep] = qk[q]; qp[keep] = qp[q]; keep++; }
        nq = keep;
    }
    CN(fm_flush)(F, A);
    /* sum_k (k+1) B_k by the running sum */
    CN(jac) run, sum;
    CN(jac_set_inf)(F, &run);
    CN(jac_set_inf)(F, &sum);
    for (size_t k = nbk; k-- > 0;) {
        if (!CN(aff_is_inf)(&A->B[k])) CN(jac_madd)(F, &run, &run, &A->B[k], 0);
        CN(jac_add)(F, &sum, &sum, &run);
    }
    J->part[t] = sum;
}

/* window width for `threads` workers: the tasks (windows x chunks) should be a few per worker, and a task's bucket reduction
 * (2^(c-1) buckets x ~27 products) must not outweigh its additions (points x ~6.5 products) */
static int CN(fm_choose)(size_t n, int scalar_bits, int threads, int* nchunk_out) {
    double best = 1e300; int best_c = 8, best_s = 1;
    for (int c = 4; c <= 16; c++) {
        const int nwin = (scalar_bits + 1 + c - 1) / c;
        int s = (2 * threads + nwin - 1) / nwin;
        if (s < 1) s = 1;
        const double per_task = (double)n / s * 6.5 + (double)((size_t)1 << (c - 1)) * 27.0;
        const double rounds = (double)((nwin * s + threads - 1) / threads);
        const double cost = per_task * rounds;
        if (cost < best) { best = cost; best_c = c; best_s = s; }
    }
    *nchunk_out = best_s;
    return best_c;
}

static void CN(fmsm)(const FPN(field) * F, const CN(aff) * pts, const uint64_t* plain_scalars, size_t n, int scalar_bits,
                     fp_pool* pool, int threads, CN(aff) * out) {
    int nchunk = 1;
    const int c = CN(fm_choose)(n, scalar_bits, threads, &nchunk);
    const int nwin = (scalar_bits + 1 + c - 1) / c;       /* one more bit: the top digit's carry */
    int32_t* dig = (int32_t*)malloc(n * (size_t)nwin * sizeof(int32_t) + 16);
    {
        const int parts = threads * 2 > 1 ? threads * 2 : 1;
        CN(fm_dig_job) D = {plain_scalars, dig, n, c, nwin, (n + parts - 1) / parts};
        if (D.per < 1) D.per = 1;
        fp_pool_run(pool, CN(fm_dig_task), &D, (int)((n + D.per - 1) / D.per));
    }
    CN(fm_job) J = {F, pts, dig, n, c, nwin, nchunk, NULL};
    J.part = (CN(jac)*)malloc((size_t)nwin * nchunk * sizeof(CN(jac)));
    fp_pool_run(pool, CN(fm_task), &J, nwin * nchunk);
    free(dig);
    CN(jac) acc;
    CN(jac_set_inf)(F, &acc);
    for (int w = nwin - 1; w >= 0; w--) {
        for (int k = 0; k < c; k++) CN(jac_dbl)(F, &acc, &acc);
        for (int ch = 0; ch < nchunk; ch++) CN(jac_add)(F, &acc, &acc, &J.part[w * nchunk + ch]);
    }
    free(J.part);
    CN(jac_to_aff)(F, out, &acc);
}
