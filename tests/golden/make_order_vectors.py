#!/usr/bin/env python3
"""Generate tests/golden/order_vectors.json: SpMV rows with INEXACT products whose expected bits are obtained by
evaluating the reference's instruction sequences in exact rational arithmetic with one IEEE rounding per operation.

The reference's own SpMV vectors have exact small-integer products and cannot tell the scalar / 4-lane / 8-lane summation
orders apart; the SpMV kernels live in headers that need AOCL-Utils and cannot be compiled here.
So each x86 intrinsic the kernels use is modelled below by its architectural definition (Intel SDM) on python Fractions,
and the kernels' statement sequences are followed line by line:

  kid 0    ref_csrmv_gn                         library/src/level2/aoclsparse_csrmv_kr.hpp:448-513
  kid 1/2  aoclsparse_csrmv_vectorized_avx2     library/src/level2/aoclsparse_csrmv_kr.hpp:949-1040
  kid 3    aoclsparse_csrmv_vectorized_avx512   library/src/level2/aoclsparse_csrmv_avx512.cpp:36-134
  float    aoclsparse_csrmv_vectorized<float>   library/src/level2/aoclsparse_csrmv_kr.hpp:734-831

This is an evaluator of the instruction stream, independent of oracle.c (no shared code, different arithmetic: exact
rationals + explicit rounding instead of hardware floating point).  The scalar statements "result += a*b" are evaluated both
ways a compiler may build them (oracle.c header): "fused" (one rounding; clang / AOCC) and "gcc_znver2" (two roundings).

Run:  python tests/golden/make_order_vectors.py
"""
import json
import os
from fractions import Fraction

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


class Fmt:
    """binary64 / binary32 round-to-nearest-even of an exact rational (normal range only; inputs are kept there)"""

    def __init__(self, dtype):
        self.dtype = dtype
        self.p = 53 if dtype == np.float64 else 24

    def rnd(self, q):
        if q == 0:
            return Fraction(0)
        s = -1 if q < 0 else 1
        q = abs(q)
        e = q.numerator.bit_length() - q.denominator.bit_length()
        if Fraction(2) ** e > q:
            e -= 1
        # 2^e <= q < 2^(e+1); ulp = 2^(e-p+1)
        ulp = Fraction(2) ** (e - self.p + 1)
        n = q / ulp
        f = n.numerator // n.denominator
        r = n - f
        if r > Fraction(1, 2) or (r == Fraction(1, 2) and f % 2 == 1):
            f += 1
        return s * f * ulp

    def add(self, a, b):
        return self.rnd(a + b)

    def mul(self, a, b):
        return self.rnd(a * b)

    def fma(self, a, b, c):
        return self.rnd(a * b + c)


# ---- intrinsic models (lane 0 first) ---------------------------------------------------------------------------------
def mm_fmadd(F, a, b, c):                 # _mm256_fmadd_pd/_ps, _mm512_fmadd_pd: lane-wise fused multiply-add
    return [F.fma(x, y, z) for x, y, z in zip(a, b, c)]


def mm256_hadd_pd(F, a, b):               # VHADDPD ymm: (a0+a1, b0+b1, a2+a3, b2+b3)
    return [F.add(a[0], a[1]), F.add(b[0], b[1]), F.add(a[2], a[3]), F.add(b[2], b[3])]


def lo128(v):                             # _mm256_castpd256_pd128 / _mm256_castps256_ps128
    return v[:len(v) // 2]


def hi128(v):                             # _mm256_extractf128_pd/_ps(v, 1)
    return v[len(v) // 2:]


def mm_add(F, a, b):                      # _mm_add_pd / _mm_add_ps / _mm256_add_pd
    return [F.add(x, y) for x, y in zip(a, b)]


def mm_movehl_ps(a, b):                   # MOVHLPS: (b2, b3, a2, a3)
    return [b[2], b[3], a[2], a[3]]


def mm_shuffle_ps_0x1(a, b):              # _mm_shuffle_ps(a, b, 0x1): (a1, a0, b0, b0)
    return [a[1], a[0], b[0], b[0]]


def scalar_acc(F, fused, a, b, acc):      # "result += a*b": one or two roundings (see module docstring)
    return F.fma(a, b, acc) if fused else F.add(F.mul(a, b), acc)


# ---- kernels: the statement sequences ---------------------------------------------------------------------------------
def row_ref(F, fused, val, xg):           # csrmv_kr.hpp:491-496
    r = Fraction(0)
    for a, x in zip(val, xg):
        r = scalar_acc(F, fused, a, x, r)
    return r


def row_avx2_d(F, fused, val, xg):        # csrmv_kr.hpp:974-1024
    n = len(val)
    k_iter, k_rem = n // 4, n % 4
    vec_y = [Fraction(0)] * 4
    for j in range(0, n - k_rem, 4):
        vec_y = mm_fmadd(F, val[j:j + 4], xg[j:j + 4], vec_y)           # :993
    result = Fraction(0)
    if k_iter:
        vec_y = mm256_hadd_pd(F, vec_y, vec_y)                           # :1000
        sse_sum = mm_add(F, lo128(vec_y), hi128(vec_y))                  # :1002-1006
        result = sse_sum[0]                                              # :1015
    for j in range(n - k_rem, n):
        result = scalar_acc(F, fused, val[j], xg[j], result)             # :1020-1023
    return result


def row_avx512_d(F, fused, val, xg):      # csrmv_avx512.cpp:58-117
    n = len(val)
    k_iter, k_rem = n // 8, n % 8
    v512 = [Fraction(0)] * 8
    for j in range(0, n - k_rem, 8):
        v512 = mm_fmadd(F, val[j:j + 8], xg[j:j + 8], v512)             # :84
    vec_y = mm_add(F, v512[:4], v512[4:])                                # :86-87 extractf64x4 0 + 1
    result = Fraction(0)
    if k_iter:
        vec_y = mm256_hadd_pd(F, vec_y, vec_y)                           # :93
        sse_sum = mm_add(F, lo128(vec_y), hi128(vec_y))                  # :95-100
        result = sse_sum[0]                                              # :110
    for j in range(n - k_rem, n):
        result = scalar_acc(F, fused, val[j], xg[j], result)             # :114-117
    return result


def row_avx2_s(F, fused, val, xg):        # csrmv_kr.hpp:760-817
    n = len(val)
    k_iter, k_rem = n // 8, n % 8
    vec_y = [Fraction(0)] * 8
    for j in range(0, n - k_rem, 8):
        vec_y = mm_fmadd(F, val[j:j + 8], xg[j:j + 8], vec_y)           # :783
    result = Fraction(0)
    if k_iter:
        hiQuad, loQuad = hi128(vec_y), lo128(vec_y)                      # :790-792
        sumQuad = mm_add(F, loQuad, hiQuad)                              # :794
        hiDual = mm_movehl_ps(sumQuad, sumQuad)                          # :798
        sumDual = mm_add(F, sumQuad, hiDual)                             # :800
        hi = mm_shuffle_ps_0x1(sumDual, sumDual)                         # :804
        result = F.add(sumDual[0], hi[0])                                # :806-807 _mm_add_ss, _mm_cvtss_f32
    for j in range(n - k_rem, n):
        result = scalar_acc(F, fused, val[j], xg[j], result)             # :811-814
    return result


def finish(F, r, alpha, beta, y):         # csrmv_kr.hpp:497-509 (same text in the three vector kernels)
    if alpha != 1:
        r = F.mul(alpha, r)
    if beta != 0:
        r = F.fma(beta, y, r)             # "result += beta * y[i]": not a loop-carried chain, fused by gcc and clang alike
    return r


def hexbits(q, dtype):
    v = np.array([float(q)], dtype=dtype)  # q is exactly representable: float() is exact, the cast too
    assert Fraction(float(v[0])) == q
    return ("%016x" % v.view(np.uint64)[0]) if dtype == np.float64 else ("%08x" % v.view(np.uint32)[0])


def main():
    rng = np.random.default_rng(4242)
    out = {"_about": "SpMV rows with inexact products; expected bits = the reference's instruction sequences evaluated in "
                     "exact rational arithmetic with one rounding per IEEE operation (tests/golden/make_order_vectors.py)",
           "rows": []}
    for dtype, name, kernels in ((np.float64, "d", (("ref", row_ref), ("avx2", row_avx2_d), ("avx512", row_avx512_d))),
                                 (np.float32, "s", (("ref", row_ref), ("avx2", row_avx2_s)))):
        F = Fmt(dtype)
        for n in list(range(0, 20)) + [31, 32, 37]:
            val = (rng.standard_normal(n) * 2.0 ** rng.integers(-4, 5, n)).astype(dtype)
            xg = (rng.standard_normal(n) * 2.0 ** rng.integers(-4, 5, n)).astype(dtype)
            ab = [(1.0, 0.0)] + ([(float(dtype(rng.standard_normal())), float(dtype(rng.standard_normal())))] if n % 3 == 1 else [])
            y0 = float(dtype(rng.standard_normal()))
            fv, fx = [Fraction(float(v)) for v in val], [Fraction(float(v)) for v in xg]
            for alpha, beta in ab:
                exp = {}
                for kname, fn in kernels:
                    for fused in (True, False):
                        r = finish(F, fn(F, fused, fv, fx), Fraction(alpha), Fraction(beta), Fraction(y0))
                        exp["%s/%s" % (kname, "fused" if fused else "gcc_znver2")] = hexbits(r, dtype)
                out["rows"].append(dict(type=name, n=n, val=[hexbits(Fraction(float(v)), dtype) for v in val],
                                        x=[hexbits(Fraction(float(v)), dtype) for v in xg], alpha=alpha, beta=beta, y0=y0,
                                        expect=exp))
    path = os.path.join(HERE, "order_vectors.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes,", len(out["rows"]), "rows")


if __name__ == "__main__":
    main()
