#!/usr/bin/env python3
"""Emit algoplonk_amd/csrc/ffu_asm.h: the unsaturated-limb Montgomery product and square of ffu.h as ONE inline-asm statement
each, for the 9 x 29-bit fields (BN254 Fp: the MSM's bucket accumulation; the two scalar fields: NTT tiles, quotient kernel)
and - round 3 - the 14 x 28-bit BLS12-381 Fp (428 instead of ~470 compiler-scheduled VALU instructions per product).

Why asm: hipcc splits every column of the C++ product scanning into two accumulator chains for instruction-level
parallelism and merges them with a v_lshl_add_u64 - 16 extra instructions per product.  tools/ubench/valu_rates.hip shows that
on gfx950 a DEPENDENT v_mad_u64_u32 chain issues exactly as fast as independent ones (2.7 ns per instruction with one wave per
SIMD, either way), so the parallelism buys nothing: one strict chain per product is 206 VALU instructions instead of 220.
An empty asm barrier after every mad gets the same chain from the compiler but makes it pad each barrier with an s_nop.

Layout of one product (L = 9 limbs, B = 29 bits, p = modulus, q = -p^-1 mod 2^B):
    column k < L : acc += sum_{i<=k} a_i b_{k-i} + sum_{i<k} m_i p_{k-i};  m_k = (lo(acc) * q) & MASK;  acc += m_k p_0;  acc >>= B
    column k >= L: acc += sum_{i>k-L} a_i b_{k-i} + m_i p_{k-i};  r_{k-L} = lo(acc) & MASK;  acc >>= B;    r_{L-1} = lo(acc)
The accumulator lives in a fixed VGPR pair (its halves must be addressable by name), the modulus in SGPRs loaded by s_mov
at the head of the statement (VOP3 takes no literals on gfx9-family targets); m_k shares its register with r_k (m_k is last
read in column k + L - 1, r_k is written in column k + L).
Run: python tools/gen_umul_asm.py > algoplonk_amd/csrc/ffu_asm.h
"""
FIELDS = {   # name: (modulus, limbs, bits per limb)
    "FpBN254": (21888242871839275222246405745257275088696311157297823662689037894645226208583, 9, 29),
    "FrBN254": (21888242871839275222246405745257275088548364400416034343698204186575808495617, 9, 29),
    "FrBLS12381": (52435875175126190479447740508185965837690552500527637822603658699938581184513, 9, 29),
    # 14 x 28 bits: 14 outputs + 28 inputs in one statement (clang has no 30-operand limit; GCC's does not apply to hipcc)
    "FpBLS12381": (4002409555221667393417789825735904156556882819939007885332058136124031650490837864442687629129015664037894272559787, 14, 28),
}
L, B, MASK = 9, 29, (1 << 29) - 1        # set per field by use()
ACC_LO, ACC_HI, TMP = 60, 61, 62          # fixed VGPRs: accumulator pair (even aligned), scratch
D0 = 64                                   # v64..v(63+L): doubled limbs of the squaring
S0 = 84                                   # s84..s(83+L) = p_0..p_(L-1), s(84+L) = q  (L = 14: s84..s98)


def use(l, b):
    global L, B, MASK
    L, B, MASK = l, b, (1 << b) - 1


def limbs(x):
    return [(x >> (B * i)) & MASK for i in range(L)]


def head(p):
    pl = limbs(p)
    q = (-pow(p, -1, 1 << B)) % (1 << B)
    out = ["s_mov_b32 s%d, 0x%08x" % (S0 + i, pl[i]) for i in range(L)]
    out.append("s_mov_b32 s%d, 0x%08x" % (S0 + L, q))
    return out


def mad(x, y, first=False):
    return "v_mad_u64_u32 v[%d:%d], vcc, %s, %s, %s" % (ACC_LO, ACC_HI, x, y, "0" if first else "v[%d:%d]" % (ACC_LO, ACC_HI))


def finish_low(k, rk):
    return ["v_mul_lo_u32 v%d, v%d, s%d" % (TMP, ACC_LO, S0 + L), "v_and_b32 %s, 0x%x, v%d" % (rk, MASK, TMP), mad(rk, "s%d" % S0),
            "v_lshrrev_b64 v[%d:%d], %d, v[%d:%d]" % (ACC_LO, ACC_HI, B, ACC_LO, ACC_HI)]


def finish_high(k, rk):
    if k == 2 * L - 2:
        return ["v_and_b32 %s, 0x%x, v%d" % (rk, MASK, ACC_LO), "v_lshrrev_b64 v[%d:%d], %d, v[%d:%d]" % (ACC_LO, ACC_HI, B, ACC_LO, ACC_HI)]
    return ["v_and_b32 %s, 0x%x, v%d" % (rk, MASK, ACC_LO), "v_lshrrev_b64 v[%d:%d], %d, v[%d:%d]" % (ACC_LO, ACC_HI, B, ACC_LO, ACC_HI)]


def body_mul(p):
    R = lambda i: "%%%d" % i                 # outputs 0..8
    A = lambda i: "%%%d" % (L + i)           # inputs
    Bq = lambda i: "%%%d" % (2 * L + i)
    ins = head(p)
    first = True
    for k in range(L):
        for i in range(k + 1):
            ins.append(mad(A(i), Bq(k - i), first)); first = False
        for i in range(k):
            ins.append(mad(R(i), "s%d" % (S0 + k - i)))
        ins += finish_low(k, R(k))
    for k in range(L, 2 * L - 1):
        for i in range(k - L + 1, L):
            ins.append(mad(A(i), Bq(k - i)))
        for i in range(k - L + 1, L):
            ins.append(mad(R(i), "s%d" % (S0 + k - i)))
        # r_{k-L} overwrites m_{k-L}, whose last use was column k-1
        ins += finish_high(k, R(k - L))
    ins.append("v_mov_b32 %s, v%d" % (R(L - 1), ACC_LO))
    return ins


def body_sqr(p):
    R = lambda i: "%%%d" % i
    A = lambda i: "%%%d" % (L + i)
    D = lambda i: "v%d" % (D0 + i)
    ins = head(p) + ["v_lshlrev_b32 %s, 1, %s" % (D(i), A(i)) for i in range(L)]
    first = True
    for k in range(L):
        for i in range((k + 1) // 2):          # 2 i < k
            ins.append(mad(D(i), A(k - i), first)); first = False
        if k % 2 == 0:
            ins.append(mad(A(k // 2), A(k // 2), first)); first = False
        for i in range(k):
            ins.append(mad(R(i), "s%d" % (S0 + k - i)))
        ins += finish_low(k, R(k))
    for k in range(L, 2 * L - 1):
        for i in range(k - L + 1, L):
            if 2 * i < k:
                ins.append(mad(D(i), A(k - i)))
        if k % 2 == 0:
            ins.append(mad(A(k // 2), A(k // 2)))
        for i in range(k - L + 1, L):
            ins.append(mad(R(i), "s%d" % (S0 + k - i)))
        ins += finish_high(k, R(k - L))
    ins.append("v_mov_b32 %s, v%d" % (R(L - 1), ACC_LO))
    return ins


def emit_fn(name, ins, nin):
    outs = ", ".join('"=&v"(r[%d])' % i for i in range(L))
    if nin == 2:
        inputs = ", ".join('"v"(a[%d])' % i for i in range(L)) + ", " + ", ".join('"v"(b[%d])' % i for i in range(L))
        sig = "uint32_t* __restrict__ r, const uint32_t* __restrict__ a, const uint32_t* __restrict__ b"
    else:
        inputs = ", ".join('"v"(a[%d])' % i for i in range(L))
        sig = "uint32_t* __restrict__ r, const uint32_t* __restrict__ a"
    clob = ['"vcc"', '"v%d"' % ACC_LO, '"v%d"' % ACC_HI, '"v%d"' % TMP] + ['"s%d"' % (S0 + i) for i in range(L + 1)]
    if nin == 1:
        clob += ['"v%d"' % (D0 + i) for i in range(L)]
    text = "\\n\\t".join(ins)
    nvalu = sum(1 for x in ins if x.startswith("v_"))
    print("    // %d VALU instructions (%d v_mad_u64_u32)" % (nvalu, sum(1 for x in ins if x.startswith("v_mad"))))
    print("    __device__ __forceinline__ static void %s(%s) {" % (name, sig))
    print('        asm("%s"\n            : %s\n            : %s\n            : %s);' % (text, outs, inputs, ", ".join(clob)))
    print("    }")


print("// GENERATED by tools/gen_umul_asm.py - do not edit.\n#pragma once\n#include <stdint.h>\n")
print("template <class P> struct UMulAsm { static constexpr bool available = false; };\n")
print("#if defined(__HIP_DEVICE_COMPILE__) && !defined(APK_NO_UMUL_ASM)")
for name, (p, l, b) in FIELDS.items():
    use(l, b)
    print("template <> struct UMulAsm<%s> {" % name)
    print("    static constexpr bool available = true;")
    emit_fn("mul", body_mul(p), 2)
    emit_fn("sqr", body_sqr(p), 1)
    print("};\n")
print("#endif")
