/* ORACLE (test infrastructure, never linked into libapk): Montgomery field arithmetic on 64-bit limbs,
 * instantiated by including this file with NL (limb count) and FN(name) (symbol prefix) defined.
 * Independent of the product's 32-bit-limb HIP templates on purpose.
 * Restates gnark-crypto v0.20.1 ecc/<curve>/{fr,fp} element arithmetic [UPSTREAM, not vendored], reached from
 * /root/reference/algoplonk.go:89.  Same R = 2^(64*NL), so byte layouts equal gnark's fr.Element / fp.Element. */

typedef struct { uint64_t l[NL]; } FN(t);

typedef struct {
    FN(t) mod, one, r2;
    uint64_t inv; /* -p^-1 mod 2^64 */
} FN(field);

static inline int FN(is_zero)(const FN(t) * a) {
    uint64_t acc = 0;
    for (int i = 0; i < NL; i++) acc |= a->l[i];
    return acc == 0;
}
static inline int FN(eq)(const FN(t) * a, const FN(t) * b) {
    uint64_t acc = 0;
    for (int i = 0; i < NL; i++) acc |= a->l[i] ^ b->l[i];
    return acc == 0;
}
static inline int FN(geq)(const FN(t) * a, const FN(t) * b) {
    for (int i = NL - 1; i >= 0; i--) {
        if (a->l[i] != b->l[i]) return a->l[i] > b->l[i];
    }
    return 1;
}
static inline uint64_t FN(sub_raw)(FN(t) * r, const FN(t) * a, const FN(t) * b) {
    unsigned __int128 borrow = 0;
    for (int i = 0; i < NL; i++) {
        unsigned __int128 t = (unsigned __int128)a->l[i] - b->l[i] - borrow;
        r->l[i] = (uint64_t)t;
        borrow = (t >> 64) & 1;
    }
    return (uint64_t)borrow;
}
static inline uint64_t FN(add_raw)(FN(t) * r, const FN(t) * a, const FN(t) * b) {
    unsigned __int128 carry = 0;
    for (int i = 0; i < NL; i++) {
        unsigned __int128 t = (unsigned __int128)a->l[i] + b->l[i] + carry;
        r->l[i] = (uint64_t)t;
        carry = t >> 64;
    }
    return (uint64_t)carry;
}
static inline void FN(add)(const FN(field) * F, FN(t) * r, const FN(t) * a, const FN(t) * b) {
    FN(t) s;
    FN(add_raw)(&s, a, b);
    if (FN(geq)(&s, &F->mod)) FN(sub_raw)(&s, &s, &F->mod);
    *r = s;
}
static inline void FN(sub)(const FN(field) * F, FN(t) * r, const FN(t) * a, const FN(t) * b) {
    FN(t) d;
    if (FN(sub_raw)(&d, a, b)) FN(add_raw)(&d, &d, &F->mod);
    *r = d;
}
static inline void FN(neg)(const FN(field) * F, FN(t) * r, const FN(t) * a) {
    if (FN(is_zero)(a)) { *r = *a; return; }
    FN(sub_raw)(r, &F->mod, a);
}
static inline void FN(dbl)(const FN(field) * F, FN(t) * r, const FN(t) * a) { FN(add)(F, r, a, a); }

/* CIOS Montgomery product; one extra word keeps the transient carry (no spare-bit assumption here). */
static inline void FN(mul)(const FN(field) * F, FN(t) * r, const FN(t) * a, const FN(t) * b) {
    uint64_t t[NL + 2];
    for (int i = 0; i < NL + 2; i++) t[i] = 0;
    for (int i = 0; i < NL; i++) {
        unsigned __int128 c = 0;
        for (int j = 0; j < NL; j++) {
            c += (unsigned __int128)a->l[j] * b->l[i] + t[j];
            t[j] = (uint64_t)c;
            c >>= 64;
        }
        c += t[NL];
        t[NL] = (uint64_t)c;
        t[NL + 1] = (uint64_t)(c >> 64);
        uint64_t m = t[0] * F->inv;
        c = (unsigned __int128)m * F->mod.l[0] + t[0];
        c >>= 64;
        for (int j = 1; j < NL; j++) {
            c += (unsigned __int128)m * F->mod.l[j] + t[j];
            t[j - 1] = (uint64_t)c;
            c >>= 64;
        }
        c += t[NL];
        t[NL - 1] = (uint64_t)c;
        t[NL] = t[NL + 1] + (uint64_t)(c >> 64);
    }
    FN(t) out;
    for (int i = 0; i < NL; i++) out.l[i] = t[i];
    if (t[NL] || FN(geq)(&out, &F->mod)) FN(sub_raw)(&out, &out, &F->mod);
    *r = out;
}
static inline void FN(sqr)(const FN(field) * F, FN(t) * r, const FN(t) * a) { FN(mul)(F, r, a, a); }

static inline void FN(from_mont)(const FN(field) * F, FN(t) * r, const FN(t) * a) {
    FN(t) o;
    for (int i = 0; i < NL; i++) o.l[i] = 0;
    o.l[0] = 1;
    FN(mul)(F, r, a, &o);
}
static inline void FN(to_mont)(const FN(field) * F, FN(t) * r, const FN(t) * a) { FN(mul)(F, r, a, &F->r2); }

/* a^e, e given as NL-limb little-endian plain integer */
static void FN(pow)(const FN(field) * F, FN(t) * r, const FN(t) * a, const uint64_t* e, int words) {
    FN(t) acc = F->one;
    for (int w = words - 1; w >= 0; w--)
        for (int b = 63; b >= 0; b--) {
            FN(sqr)(F, &acc, &acc);
            if ((e[w] >> b) & 1) FN(mul)(F, &acc, &acc, a);
        }
    *r = acc;
}
static void FN(pow_u64)(const FN(field) * F, FN(t) * r, const FN(t) * a, uint64_t e) { FN(pow)(F, r, a, &e, 1); }
/* Fermat inverse; inv(0) = 0 */
static void FN(inv)(const FN(field) * F, FN(t) * r, const FN(t) * a) {
    FN(t) e = F->mod;
    FN(t) two;
    for (int i = 0; i < NL; i++) two.l[i] = 0;
    two.l[0] = 2;
    FN(sub_raw)(&e, &e, &two);
    FN(pow)(F, r, a, e.l, NL);
}
