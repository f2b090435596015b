/* ORACLE (test infrastructure): G1 arithmetic + Pippenger MSM, instantiated per base field by defining
 * FPN(name) (base-field prefix, from field_tmpl.h) and CN(name) (curve prefix).
 * Restates gnark-crypto v0.20.1 ecc/<curve>/g1.go + multiexp.go [UPSTREAM, not vendored; SURVEY.md §3.4]:
 * Jacobian coordinates, mixed addition, one bucket set PER WINDOW with the running-sum reduction and a final
 * double-and-add over windows - the classic CPU algorithm, deliberately not the fixed-base single-bucket-set
 * design of the HIP path. */

typedef struct { FPN(t) x, y; } CN(aff);           /* (0,0) = infinity, gnark layout */
typedef struct { FPN(t) X, Y, Z; } CN(jac);        /* Z == 0 = infinity */

static inline int CN(aff_is_inf)(const CN(aff) * p) { return FPN(is_zero)(&p->x) && FPN(is_zero)(&p->y); }
static inline void CN(jac_set_inf)(const FPN(field) * F, CN(jac) * p) { p->X = F->one; p->Y = F->one; memset(&p->Z, 0, sizeof p->Z); }
static inline int CN(jac_is_inf)(const CN(jac) * p) { return FPN(is_zero)(&p->Z); }

/* dbl-2009-l (a = 0) */
static void CN(jac_dbl)(const FPN(field) * F, CN(jac) * r, const CN(jac) * p) {
    if (CN(jac_is_inf)(p)) { *r = *p; return; }
    FPN(t) A, B, C, D, E, Fq, t;
    FPN(sqr)(F, &A, &p->X);
    FPN(sqr)(F, &B, &p->Y);
    FPN(sqr)(F, &C, &B);
    FPN(add)(F, &t, &p->X, &B);
    FPN(sqr)(F, &t, &t);
    FPN(sub)(F, &t, &t, &A);
    FPN(sub)(F, &t, &t, &C);
    FPN(dbl)(F, &D, &t);
    FPN(dbl)(F, &E, &A);
    FPN(add)(F, &E, &E, &A);
    FPN(sqr)(F, &Fq, &E);
    FPN(t) X3, Y3, Z3;
    FPN(dbl)(F, &t, &D);
    FPN(sub)(F, &X3, &Fq, &t);
    FPN(sub)(F, &t, &D, &X3);
    FPN(mul)(F, &Y3, &E, &t);
    FPN(dbl)(F, &t, &C); FPN(dbl)(F, &t, &t); FPN(dbl)(F, &t, &t);
    FPN(sub)(F, &Y3, &Y3, &t);
    FPN(mul)(F, &Z3, &p->Y, &p->Z);
    FPN(dbl)(F, &Z3, &Z3);
    r->X = X3; r->Y = Y3; r->Z = Z3;
}

/* madd-2007-bl: r = p + q (q affine, optionally negated) */
static void CN(jac_madd)(const FPN(field) * F, CN(jac) * r, const CN(jac) * p, const CN(aff) * q_in, int negate) {
    if (CN(aff_is_inf)(q_in)) { *r = *p; return; }
    CN(aff) q = *q_in;
    if (negate) FPN(neg)(F, &q.y, &q.y);
    if (CN(jac_is_inf)(p)) { r->X = q.x; r->Y = q.y; r->Z = F->one; return; }
    FPN(t) Z1Z1, U2, S2, H, HH, I, J, rr, V, t;
    FPN(sqr)(F, &Z1Z1, &p->Z);
    FPN(mul)(F, &U2, &q.x, &Z1Z1);
    FPN(mul)(F, &S2, &q.y, &p->Z);
    FPN(mul)(F, &S2, &S2, &Z1Z1);
    if (FPN(eq)(&U2, &p->X)) {
        if (FPN(eq)(&S2, &p->Y)) { CN(jac_dbl)(F, r, p); return; }
        CN(jac_set_inf)(F, r);
        return;
    }
    FPN(sub)(F, &H, &U2, &p->X);
    FPN(sqr)(F, &HH, &H);
    FPN(dbl)(F, &I, &HH); FPN(dbl)(F, &I, &I);
    FPN(mul)(F, &J, &H, &I);
    FPN(sub)(F, &rr, &S2, &p->Y);
    FPN(dbl)(F, &rr, &rr);
    FPN(mul)(F, &V, &p->X, &I);
    FPN(t) X3, Y3, Z3;
    FPN(sqr)(F, &X3, &rr);
    FPN(sub)(F, &X3, &X3, &J);
    FPN(sub)(F, &X3, &X3, &V);
    FPN(sub)(F, &X3, &X3, &V);
    FPN(sub)(F, &t, &V, &X3);
    FPN(mul)(F, &Y3, &rr, &t);
    FPN(mul)(F, &t, &p->Y, &J);
    FPN(dbl)(F, &t, &t);
    FPN(sub)(F, &Y3, &Y3, &t);
    FPN(add)(F, &Z3, &p->Z, &H);
    FPN(sqr)(F, &Z3, &Z3);
    FPN(sub)(F, &Z3, &Z3, &Z1Z1);
    FPN(sub)(F, &Z3, &Z3, &HH);
    r->X = X3; r->Y = Y3; r->Z = Z3;
}

/* add-2007-bl */
static void CN(jac_add)(const FPN(field) * F, CN(jac) * r, const CN(jac) * p, const CN(jac) * q) {
    if (CN(jac_is_inf)(q)) { *r = *p; return; }
    if (CN(jac_is_inf)(p)) { *r = *q; return; }
    FPN(t) Z1Z1, Z2Z2, U1, U2, S1, S2, H, I, J, rr, V, t;
    FPN(sqr)(F, &Z1Z1, &p->Z);
    FPN(sqr)(F, &Z2Z2, &q->Z);
    FPN(mul)(F, &U1, &p->X, &Z2Z2);
    FPN(mul)(F, &U2, &q->X, &Z1Z1);
    FPN(mul)(F, &S1, &p->Y, &q->Z); FPN(mul)(F, &S1, &S1, &Z2Z2);
    FPN(mul)(F, &S2, &q->Y, &p->Z); FPN(mul)(F, &S2, &S2, &Z1Z1);
    if (FPN(eq)(&U1, &U2)) {
        if (FPN(eq)(&S1, &S2)) { CN(jac_dbl)(F, r, p); return; }
        CN(jac_set_inf)(F, r);
        return;
    }
    FPN(sub)(F, &H, &U2, &U1);
    FPN(dbl)(F, &I, &H); FPN(sqr)(F, &I, &I);
    FPN(mul)(F, &J, &H, &I);
    FPN(sub)(F, &rr, &S2, &S1); FPN(dbl)(F, &rr, &rr);
    FPN(mul)(F, &V, &U1, &I);
    FPN(t) X3, Y3, Z3;
    FPN(sqr)(F, &X3, &rr);
    FPN(sub)(F, &X3, &X3, &J); FPN(sub)(F, &X3, &X3, &V); FPN(sub)(F, &X3, &X3, &V);
    FPN(sub)(F, &t, &V, &X3);
    FPN(mul)(F, &Y3, &rr, &t);
    FPN(mul)(F, &t, &S1, &J); FPN(dbl)(F, &t, &t);
    FPN(sub)(F, &Y3, &Y3, &t);
    FPN(add)(F, &Z3, &p->Z, &q->Z); FPN(sqr)(F, &Z3, &Z3);
    FPN(sub)(F, &Z3, &Z3, &Z1Z1); FPN(sub)(F, &Z3, &Z3, &Z2Z2);
    FPN(mul)(F, &Z3, &Z3, &H);
    r->X = X3; r->Y = Y3; r->Z = Z3;
}

static void CN(jac_to_aff)(const FPN(field) * F, CN(aff) * r, const CN(jac) * p) {
    if (CN(jac_is_inf)(p)) { memset(r, 0, sizeof *r); return; }
    FPN(t) zi, zi2;
    FPN(inv)(F, &zi, &p->Z);
    FPN(sqr)(F, &zi2, &zi);
    FPN(mul)(F, &r->x, &p->X, &zi2);
    FPN(mul)(F, &zi2, &zi2, &zi);
    FPN(mul)(F, &r->y, &p->Y, &zi2);
}

/* ---- Pippenger: tasks = windows x point-chunks (so every host core has work) ------------------------------- */
typedef struct {
    const FPN(field) * F;
    const CN(aff) * pts;
    const uint64_t* sc; /* n x 4 plain (non-Montgomery) scalar limbs */
    size_t n;
    int c, nwin, nchunk;
    CN(jac) * part; /* [nwin][nchunk] */
} CN(msm_job);

static void CN(msm_task)(void* arg, int t) {
    CN(msm_job)* J = (CN(msm_job)*)arg;
    const FPN(field)* F = J->F;
    const int c = J->c, w = t / J->nchunk, ch = t % J->nchunk;
    const size_t nb = (size_t)1 << c; /* unsigned digits: buckets 1 .. 2^c - 1 */
    const size_t per = (J->n + J->nchunk - 1) / J->nchunk;
    const size_t lo = (size_t)ch * per < J->n ? (size_t)ch * per : J->n, hi = lo + per < J->n ? lo + per : J->n;
    CN(jac)* B = (CN(jac)*)malloc(nb * sizeof(CN(jac)));
    for (size_t k = 0; k < nb; k++) CN(jac_set_inf)(F, &B[k]);
    const int bit = w * c;
    for (size_t i = lo; i < hi; i++) {
        const uint64_t* s = J->sc + 4 * i;
        int word = bit >> 6, off = bit & 63;
        uint64_t v = word < 4 ? s[word] >> off : 0;
        if (off + c > 64 && word + 1 < 4) v |= s[word + 1] << (64 - off);
        v &= nb - 1;
        if (v) CN(jac_madd)(F, &B[v], &B[v], &J->pts[i], 0);
    }
    CN(jac) run, sum;
    CN(jac_set_inf)(F, &run);
    CN(jac_set_inf)(F, &sum);
    for (size_t k = nb - 1; k >= 1; k--) {
        CN(jac_add)(F, &run, &run, &B[k]);
        CN(jac_add)(F, &sum, &sum, &run);
    }
    J->part[t] = sum;
    free(B);
}

static void CN(msm)(const FPN(field) * F, const CN(aff) * pts, const uint64_t* plain_scalars, size_t n, int scalar_bits,
                    int threads, CN(aff) * out) {
    int nchunk = 1;
    if (threads > 32 && n >= 4096) { nchunk = threads / 24; if (nchunk < 1) nchunk = 1; }
    size_t m = (n + nchunk - 1) / nchunk;
    int lg = 0;
    while (m > 1) { m >>= 1; lg++; }
    int c = lg > 8 ? lg - 4 : (lg > 3 ? lg - 1 : 2);
    if (c > 16) c = 16;
    if (c < 2) c = 2;
    int nwin = (scalar_bits + c - 1) / c;
    CN(msm_job) J = {F, pts, plain_scalars, n, c, nwin, nchunk, NULL};
    J.part = (CN(jac)*)malloc((size_t)nwin * nchunk * sizeof(CN(jac)));
    parallel_for(CN(msm_task), &J, nwin * nchunk, threads);
    CN(jac) acc;
    CN(jac_set_inf)(F, &acc);
    for (int w = nwin - 1; w >= 0; w--) {
        for (int k = 0; k < c; k++) CN(jac_dbl)(F, &acc, &acc);
        for (int ch = 0; ch < nchunk; ch++) CN(jac_add)(F, &acc, &acc, &J.part[w * nchunk + ch]);
    }
    free(J.part);
    CN(jac_to_aff)(F, out, &acc);
}
