// Setup-time kernels: what `setup.Run`'s trusted branch does on the CPU before the
// first proof - /root/reference/setup/setup.go:110-149.
//   * g1_decompress_kernel   : kzg SRS `ReadFrom` point decompression (setup/setup.go:173-174,189-190): one Fp square
//                              root per point.  Encoding = SURVEY.md App. A.5; KATs setup/trusted_setup_test.go:53-59,183-189.
//   * lagrange_*_kernel      : kzg.ToLagrangeG1 (setup/setup.go:124,138) [UPSTREAM gnark-crypto]: an inverse FFT "in the
//                              exponent" - radix-2 butterflies whose multiplications are G1 scalar multiplications by the
//                              twiddles - so that L_i = [L_i(tau)]G1 without knowing tau.
#pragma once
#include "ec.h"

namespace apk {

// k * P for a canonical (non-Montgomery) scalar k
template <class FR, class FP>
__device__ __forceinline__ XYZZ<FP> scalar_mul_point(const XYZZ<FP>& P, const Fe<FR>& k) {
    XYZZ<FP> acc = XYZZ<FP>::inf();
    bool started = false;
    for (int w = Fe<FR>::N - 1; w >= 0; w--) {
        for (int b = 31; b >= 0; b--) {
            if (started) acc = XYZZ<FP>::dbl(acc);
            if ((k.l[w] >> b) & 1u) {
                acc.add(P);
                started = true;
            }
        }
    }
    return acc;
}

// in: count x (4*N) big-endian bytes with gnark's flag bits; out: affine Montgomery; err[0] != 0 on a bad point.
// Same acceptance as gnark's kzg SRS ReadFrom -> G1Affine.SetBytes [UPSTREAM]: flags, x < p, on the curve, and - on
// BLS12-381, whose G1 has a cofactor - membership of the order-r subgroup ([r]P = infinity; setup-time only, so the plain
// 255-bit chain is fine); an infinity encoding must carry nothing but its flag.
template <class FR, class FP, int CURVE_ID>
__global__ void __launch_bounds__(256) g1_decompress_kernel(const uint8_t* __restrict__ in, uint32_t count, Affine<FP>* __restrict__ out,
                                                            uint32_t* __restrict__ err) {
    using F = Fe<FP>;
    constexpr int N = FP::N, NB = 4 * N;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint8_t* p = in + (size_t)i * NB;
    F x;
#pragma unroll
    for (int w = 0; w < N; w++) {
        const uint8_t* q = p + NB - 4 * (w + 1);
        x.l[w] = (uint32_t)q[0] << 24 | (uint32_t)q[1] << 16 | (uint32_t)q[2] << 8 | q[3];
    }
    // flags in the top bits of byte 0: BN254 2 bits (10 smallest y, 11 largest, 01 infinity), BLS12-381 3 bits
    // (100 / 101 / 110)
    const uint32_t top = p[0];
    bool inf, largest, bad = false;
    if (CURVE_ID == 0) {
        const uint32_t f = top >> 6;
        inf = f == 1; largest = f == 3; bad = f == 0;
        x.l[N - 1] &= 0x3fffffffu;
    } else {
        const uint32_t f = top >> 5;
        inf = f == 6; largest = f == 5; bad = !(f == 4 || f == 5 || f == 6);
        x.l[N - 1] &= 0x1fffffffu;
    }
    if (inf) {
        if (!x.is_zero()) atomicAdd(err, 1u);   // non-zero payload under the infinity flag
        out[i] = Affine<FP>::inf();
        return;
    }
    // canonical range check then Montgomery form
    F m = F::modulus();
    bool lt = false;
    for (int w = N - 1; w >= 0; w--) {
        if (x.l[w] != m.l[w]) { lt = x.l[w] < m.l[w]; break; }
    }
    if (!lt) bad = true;
    F xm = F::to_mont(x);
    F b = F::zero();
    b.l[0] = FP::CURVE_B;
    F rhs = F::sqr(xm) * xm + F::to_mont(b);
    uint32_t e[N];
#pragma unroll
    for (int w = 0; w < N; w++) e[w] = FP::sqrt_exp(w);
    F y = F::pow(rhs, e, N);
    if (F::sqr(y) != rhs) bad = true;
    F yc = F::from_mont(y);
    bool y_large = false;  // yc > (p-1)/2
    for (int w = N - 1; w >= 0; w--) {
        uint32_t h = FP::half(w);
        if (yc.l[w] != h) { y_large = yc.l[w] > h; break; }
    }
    if (y_large != largest) y = F::neg(y);
    if (CURVE_ID == 1 && !bad) {
        const XYZZ<FP> rp = scalar_mul_point<FR, FP>(XYZZ<FP>::from_affine(Affine<FP>{xm, y}), Fe<FR>::modulus());
        if (!rp.is_inf()) bad = true;   // a point of the curve outside G1
    }
    if (bad) { atomicAdd(err, 1u); out[i] = Affine<FP>::inf(); return; }
    out[i] = Affine<FP>{xm, y};
}

// load affine points bit-reversed into XYZZ work space
template <class FP>
__global__ void __launch_bounds__(256) lagrange_load_kernel(const Affine<FP>* __restrict__ in, uint32_t n, int log_n,
                                                            XYZZ<FP>* __restrict__ work) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t j = __brev(i) >> (32 - log_n);
    work[i] = XYZZ<FP>::from_affine(in[j]);
}

// one decimation-in-time stage t: pairs (i, i + 2^t), twiddle = winv^(j * n / 2^(t+1)), j = i mod 2^t.
// twi[] = powers of omega^-1 (Montgomery), the size-n inverse twiddle table of the circuit context
template <class FR, class FP>
__global__ void __launch_bounds__(128) lagrange_stage_kernel(XYZZ<FP>* __restrict__ work, const Fe<FR>* __restrict__ twi, uint32_t n,
                                                             int log_n, int t) {
    const uint32_t bf = blockIdx.x * blockDim.x + threadIdx.x;
    if (bf >= n / 2) return;
    const uint32_t low = bf & ((1u << t) - 1u);
    const uint32_t i0 = ((bf >> t) << (t + 1)) | low, i1 = i0 | (1u << t);
    XYZZ<FP> u = work[i0], v = work[i1];
    if (low != 0) {
        Fe<FR> w = Fe<FR>::from_mont(twi[low << (log_n - 1 - t)]);
        v = scalar_mul_point<FR, FP>(v, w);
    }
    XYZZ<FP> s = u;
    s.add(v);
    v.neg_inplace();
    u.add(v);
    work[i0] = s;
    work[i1] = u;
}

// out[i] = (1/n) * work[i] in affine form
template <class FR, class FP>
__global__ void __launch_bounds__(128) lagrange_finish_kernel(const XYZZ<FP>* __restrict__ work, uint32_t n, Fe<FR> n_inv_mont,
                                                              Affine<FP>* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = scalar_mul_point<FR, FP>(work[i], Fe<FR>::from_mont(n_inv_mont)).to_affine();
}

}  // namespace apk
