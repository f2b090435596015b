// Device check of the four-lane point operations (ec.h add_quad_general / dbl_quad_general) against the one-lane lazy forms,
// lane by lane and through a 128-element LDS tree as the reduction kernels run it.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I algoplonk_amd/csrc tools/ubench/quad_check.hip -o tools/ubench/quad_check.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include "ec.h"
using FP = FpBN254;
using PT = XYZZ<FP, FeU<FP>>;
using AffU = Affine<FP, FeU<FP>>;

__device__ AffU gen() {
    Affine<FP> g;
    for (int i = 0; i < Fe<FP>::N; i++) g.x.l[i] = FP::one(i);
    g.y = Fe<FP>::add(g.x, g.x);                              // (1, 2)
    return unpack_affine<FP>(to_table_record<FP>(g));
}
__device__ bool same(const PT& a, const PT& b) {
    Affine<FP> x = to_fe_point<FP>(a).to_affine(), y = to_fe_point<FP>(b).to_affine();
    return x.x == y.x && x.y == y.y;
}

__global__ void pairs(int* bad) {
    const int q = threadIdx.x & 3, t = threadIdx.x >> 2;
    AffU gu = gen();
    PT a = PT::from_affine(gu);
    for (int i = 0; i < t + 1; i++) a = PT::dbl_lazy(a);      // 2^(t+1) G
    PT b = PT::from_affine(gu);
    b.add_lazy(a);                                            // (2^(t+1) + 1) G
    PT r1 = a; r1.add_lazy(b);
    PT r2 = a; bool deg; r2.add_quad_general(b, q, deg);
    PT d1 = PT::dbl_lazy(b), d2 = PT::dbl_quad_general(b, q);
    bad[threadIdx.x] = (same(r1, r2) && !deg ? 0 : 1) | (same(d1, d2) ? 0 : 2);
}

__global__ void tree(int* bad, int with_inf) {
    __shared__ PT sm[128];
    const int q = threadIdx.x & 3, t = threadIdx.x >> 2;
    AffU gu = gen();
    PT p = PT::inf();
    for (int b = 7; b >= 0; b--) { p = PT::dbl_lazy(p); if (((t + 1) >> b) & 1) p.add_lazy(PT::from_affine(gu)); }   // (t + 1) G
    if (with_inf && (t % 3) == 1) p = PT::inf();
    PT acc = p;
    if (q == 0) sm[t] = acc;
    __syncthreads();
    for (int d = 64; d >= 1; d >>= 1) {
        if (t < d) {
            PT o = sm[t + d];
            if (acc.is_inf()) acc = o;
            else if (!o.is_inf()) { bool deg; acc.add_quad_general(o, q, deg); if (deg) acc.add_lazy(o); }
            if (q == 0) sm[t] = acc;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        int k = 0;
        for (int i = 0; i < 128; i++) if (!(with_inf && (i % 3) == 1)) k += i + 1;
        PT e = PT::inf();
        for (int b = 15; b >= 0; b--) { e = PT::dbl_lazy(e); if ((k >> b) & 1) e.add_lazy(PT::from_affine(gu)); }
        bad[0] = same(acc, e) ? 0 : 1;
    }
}

int main() {
    int* bad;
    if (hipMalloc(&bad, 256 * 4) != hipSuccess) return 2;
    int fails = 0;
    for (int w = 0; w < 2; w++) {
        tree<<<1, 512>>>(bad, w);
        int hb = -1;
        if (hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost) != hipSuccess) return 2;
        printf("tree reduction (with_inf=%d): %s\n", w, hb == 0 ? "ok" : "MISMATCH");
        fails += hb != 0;
    }
    pairs<<<1, 256>>>(bad);
    int h[256];
    if (hipMemcpy(h, bad, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) return 2;
    int na = 0, nd = 0;
    for (int i = 0; i < 256; i++) { na += h[i] & 1; nd += (h[i] >> 1) & 1; }
    printf("add mismatches %d / 256, dbl mismatches %d / 256\n", na, nd);
    return fails + na + nd != 0;
}
