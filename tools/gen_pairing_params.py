#!/usr/bin/env python3
"""Emit algoplonk_amd/csrc/pairing_params.h: constants of the HOST-side pairing check behind apk_verify (the library's
mirror of gnark's plonk.Verify, /root/reference/algoplonk.go:93) for BN254 and BLS12-381.  Every constant is derived or
checked numerically here, never typed into the C++.
Run: python tools/gen_pairing_params.py > algoplonk_amd/csrc/pairing_params.h

Tower: Fp2 = Fp[u]/(u^2+1), Fp6 = Fp2[v]/(v^3 - xi), Fp12 = Fp6[w]/(w^2 - v), xi = XI0 + u.
Pairing used by the check: the plain ate pairing a(Q, P) = f_{T,Q}(P)^((p^12-1)/r) with T = |t - 1| (t = trace of Frobenius):
no Frobenius maps anywhere; a negative T only inverts the value, which a product-equals-one check does not see.
"""
P_BN = 21888242871839275222246405745257275088696311157297823662689037894645226208583
R_BN = 21888242871839275222246405745257275088548364400416034343698204186575808495617
X_BN = 4965661367192848881
P_BLS = 4002409555221667393417789825735904156556882819939007885332058136124031650490837864442687629129015664037894272559787
R_BLS = 52435875175126190479447740508185965837690552500527637822603658699938581184513
X_BLS = -0xd201000000010000

# G2 generators (the points gnark's kzg SRS carries as G2[0]; tests/golden/*.vk.bin decode to them: tests/test_verify_host.py)
G2_BN = ((10857046999023057135944570762232829481370756359578518086990519993285655852781,
          11559732032986387107991004021392285783925812861821192530917403151452391805634),
         (8495653923123431417604973247489272438418190587263600148770280649306958101930,
          4082367875863433681332203403145435568316851327593401208105741076214120093531))
G2_BLS = ((0x024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8,
           0x13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e),
          (0x0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801,
           0x0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be))


def f2mul(a, b, p):
    return ((a[0] * b[0] - a[1] * b[1]) % p, (a[0] * b[1] + a[1] * b[0]) % p)


def f2inv(a, p):
    n = pow(a[0] * a[0] + a[1] * a[1], -1, p)
    return (a[0] * n % p, -a[1] * n % p)


def f2add(a, b, p):
    return ((a[0] + b[0]) % p, (a[1] + b[1]) % p)


def f2sub(a, b, p):
    return ((a[0] - b[0]) % p, (a[1] - b[1]) % p)


def g2_add(A, B, p):
    if A is None:
        return B
    if B is None:
        return A
    if A[0] == B[0]:
        if f2add(A[1], B[1], p) == (0, 0):
            return None
        lam = f2mul(f2mul((3, 0), f2mul(A[0], A[0], p), p), f2inv(f2mul((2, 0), A[1], p), p), p)
    else:
        lam = f2mul(f2sub(B[1], A[1], p), f2inv(f2sub(B[0], A[0], p), p), p)
    x = f2sub(f2sub(f2mul(lam, lam, p), A[0], p), B[0], p)
    return (x, f2sub(f2mul(lam, f2sub(A[0], x, p), p), A[1], p))


def g2_mul(A, k, p):
    acc = None
    while k:
        if k & 1:
            acc = g2_add(acc, A, p)
        A = g2_add(A, A, p)
        k >>= 1
    return acc


def limbs(x, n):
    return [(x >> (32 * i)) & 0xFFFFFFFF for i in range(n)]


def arr(name, v):
    body = ", ".join("0x%08xu" % w for w in v)
    return ("    static constexpr uint32_t %s(int i) {\n"
            "        constexpr uint32_t v[%d] = {%s};\n        return v[i];\n    }\n") % (name, len(v), body)


def emit(name, p, r, x, xi0, twist_m, b, g2):
    n = (p.bit_length() + 31) // 32
    xi = (xi0, 1)
    bt = f2mul((b, 0), xi, p) if twist_m else f2mul((b, 0), f2inv(xi, p), p)
    # generator lies on the twist y^2 = x^3 + b' and has order r
    X, Y = g2
    assert f2mul(Y, Y, p) == f2add(f2mul(f2mul(X, X, p), X, p), bt, p), name + ": G2 generator not on the twist"
    assert g2_mul(g2, r, p) is None, name + ": G2 generator not of order r"
    t = (x + 1) if name == "PairBLS12381" else (6 * x * x + 1)       # trace of Frobenius
    assert (p + 1 - t) % r == 0, name + ": r does not divide #E(Fp)"
    T = abs(t - 1)
    assert (T - p) % r == 0 or (T + p) % r == 0, name + ": T != +-p mod r"
    assert (p ** 6 + 1) % r == 0
    fe = (p ** 6 + 1) // r
    few = (fe.bit_length() + 31) // 32
    tw = (T.bit_length() + 31) // 32
    print("struct %s {" % name)
    print("    static constexpr uint32_t XI0 = %d;          // xi = XI0 + u" % xi0)
    print("    static constexpr bool TWIST_M = %s;       // M-type twist (b' = b xi) or D-type (b' = b / xi)" % ("true" if twist_m else "false"))
    print("    static constexpr int ATE_BITS = %d;         // T = |t - 1|" % T.bit_length())
    print(arr("ate", limbs(T, tw)), end="")
    print("    static constexpr int FEXP_BITS = %d;      // (p^6 + 1) / r" % fe.bit_length())
    print(arr("fexp", limbs(fe, few)), end="")
    print(arr("bt0", limbs(bt[0], n)), end="")
    print(arr("bt1", limbs(bt[1], n)), end="")
    for nm, v in (("g2x0", X[0]), ("g2x1", X[1]), ("g2y0", Y[0]), ("g2y1", Y[1])):
        print(arr(nm, limbs(v, n)), end="")
    print("};\n")


print("// GENERATED by tools/gen_pairing_params.py - do not edit.  Host-only constants of apk_verify's pairing check.\n#pragma once\n#include <stdint.h>\n")
emit("PairBN254", P_BN, R_BN, X_BN, 9, False, 3, G2_BN)
emit("PairBLS12381", P_BLS, R_BLS, X_BLS, 1, True, 4, G2_BLS)
