#!/usr/bin/env python3
"""Generate tests/golden/kt_vectors.json: outputs of the reference's OWN vector micro-kernels
(library/src/include/kernel-templates/*.hpp, compiled from where they lie into oracle/_ref/libktref.so behind
oracle/ref_kt_driver.cpp) on seeded random, INEXACT inputs.

Why: the reference's unit-test vectors have exact small-integer products, so they cannot tell one summation order from
another.  These vectors can: kt_hsum_p / kt_dot_p trees for 256- and 512-bit registers, the row
sequence of kt_trsv_l / kt_trsv_u (trsv_kt.cpp:92-137) and the element / row sequences of csrmm_col_kt / csrmm_row_kt
(csrmm_kt.cpp:127-191, :244-356) -- every float is stored as its bit pattern (hex).

Run in the build container only (needs /root/reference and an AVX-512 host):
    make -C oracle ktref && python tests/golden/make_kt_vectors.py
The committed JSON is what tests/test_oracle_golden.py reads; nothing under tests/ touches /root/reference at run time.
"""
import ctypes
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
import oracle  # noqa: E402

P = ctypes.c_void_p
ci, cd, cf = ctypes.c_int, ctypes.c_double, ctypes.c_float


def p(a):
    return a.ctypes.data_as(P)


def hx(a):
    a = np.asarray(a)
    if a.dtype == np.float32:
        return ["%08x" % v for v in a.view(np.uint32).ravel()]
    return ["%016x" % v for v in np.asarray(a, np.float64).view(np.uint64).ravel()]


def rnd(rng, n, dtype=np.float64):
    """values spread over a few binades with full mantissas: products and sums all round"""
    return (rng.standard_normal(n) * 2.0 ** rng.integers(-6, 7, n)).astype(dtype)


def main():
    L = oracle.ktref()
    assert L is not None and L.ktref_have_avx512(), "needs oracle/_ref/libktref.so and an AVX-512 host"
    rng = np.random.default_rng(20261002)
    out = {"_about": "outputs of amd/aocl-sparse v5.3.2 kernel-template micro-kernels (kt_hsum_p, kt_dot_p, the row "
                     "sequences of kt_trsv_l/u, csrmm_col_kt, csrmm_row_kt) compiled from /root/reference by oracle/Makefile "
                     "(target ktref) with the reference's flags; bit patterns in hex; generator tests/golden/make_kt_vectors.py",
           "src": {"hsum": "library/src/include/kernel-templates/kt_l0_avx2.hpp:331-351, kt_l0_avx512.hpp:367-376",
                   "dot": "library/src/include/kernel-templates/kt_l1.hpp:41-46",
                   "trsv_row": "library/src/level2/aoclsparse_trsv_kt.cpp:92-137, 324-371",
                   "csrmm_col": "library/src/level3/aoclsparse_csrmm_kt.cpp:127-191",
                   "csrmm_row": "library/src/level3/aoclsparse_csrmm_kt.cpp:244-356"}}
    hs = []
    for bits, tsz, dt in ((256, 4, np.float64), (512, 8, np.float64), (256, 8, np.float32), (512, 16, np.float32)):
        for _ in range(12):
            v, w = rnd(rng, tsz, dt), rnd(rng, tsz, dt)
            if dt == np.float64:
                h, d = L.ktref_hsum_d(ci(bits), p(v)), L.ktref_dot_d(ci(bits), p(v), p(w))
                h, d = np.float64(h), np.float64(d)
            else:
                h, d = np.float32(L.ktref_hsum_s(ci(bits), p(v))), np.float32(L.ktref_dot_s(ci(bits), p(v), p(w)))
            hs.append(dict(bits=bits, tsz=tsz, type="d" if dt == np.float64 else "s", v=hx(v), w=hx(w),
                           hsum=hx(np.array([h], dt))[0], dot=hx(np.array([d], dt))[0]))
    out["hsum_dot"] = hs

    rows = []
    for dt, name in ((np.float64, "d"), (np.float32, "s")):
        for kid in (1, 2, 3):
            tsz = {("d", 1): 4, ("d", 2): 4, ("d", 3): 8, ("s", 1): 8, ("s", 2): 8, ("s", 3): 16}[(name, kid)]
            for cnt in list(range(0, 2 * tsz + 3)) + [5 * tsz - 1, 5 * tsz + 2]:
                a, x = rnd(rng, max(cnt, 1), dt), rnd(rng, cnt + 3, dt)
                icol = rng.permutation(cnt + 3)[:max(cnt, 1)].astype(np.int32)
                xi = rnd(rng, 1, dt)[0]
                if name == "d":
                    r = np.float64(L.ktref_trsv_row_d(ci(kid), cd(xi), ci(cnt), p(a), p(x), p(icol)))
                else:
                    r = np.float32(L.ktref_trsv_row_s(ci(kid), cf(xi), ci(cnt), p(a), p(x), p(icol)))
                rows.append(dict(type=name, kid=kid, tsz=tsz, cnt=cnt, xi=hx(np.array([xi], dt))[0], a=hx(a[:cnt]),
                                 x=hx(x), icol=[int(c) for c in icol[:cnt]], out=hx(np.array([r], dt))[0]))
    out["trsv_row"] = rows

    col = []
    for bits, psz in ((256, 4), (512, 8)):
        for nnz in list(range(0, 2 * psz + 3)) + [37]:
            a, b = rnd(rng, max(nnz, 1)), rnd(rng, nnz + 2)
            icol = rng.permutation(nnz + 2)[:max(nnz, 1)].astype(np.int32)
            for alpha, beta in ((1.0, 0.0), (rnd(rng, 1)[0], rnd(rng, 1)[0])):
                c0 = rnd(rng, 1)[0]
                r = np.float64(L.ktref_csrmm_col_elem_d(ci(bits), ci(nnz), p(a), p(b), p(icol), cd(alpha), cd(beta), cd(c0)))
                col.append(dict(psz=psz, nnz=nnz, a=hx(a[:nnz]), b=hx(b), icol=[int(c) for c in icol[:nnz]],
                                alpha=hx([alpha])[0], beta=hx([beta])[0], c0=hx([c0])[0], out=hx([r])[0]))
    out["csrmm_col"] = col

    row = []
    for bits, psz in ((256, 4), (512, 8)):
        for n in (1, 3, 4, 7, 8, 9, 16, 19):
            for nnz in (0, 1, 3, 4, 6, 9):
                k = nnz + 2
                a, B = rnd(rng, max(nnz, 1)), rnd(rng, k * n)
                icol = rng.permutation(k)[:max(nnz, 1)].astype(np.int32)
                alpha, beta = rnd(rng, 2)
                c = rnd(rng, n)
                c0 = c.copy()
                st = L.ktref_csrmm_row_d(ci(bits), ci(nnz), p(a), p(B), ci(n), p(icol), ci(n), cd(alpha), cd(beta), p(c))
                assert st == 0
                row.append(dict(psz=psz, n=n, nnz=nnz, k=k, a=hx(a[:nnz]), B=hx(B), icol=[int(v) for v in icol[:nnz]],
                                alpha=hx([alpha])[0], beta=hx([beta])[0], c0=hx(c0), out=hx(c)))
    out["csrmm_row"] = row

    # float csrmm (added later in round 3): its own generator, so that every vector above keeps the bits it was committed with
    rngs = np.random.default_rng(20261003)
    f32 = np.float32
    cols = []
    for bits, psz in ((256, 8), (512, 16)):
        for nnz in list(range(0, 2 * psz + 3)) + [5 * psz + 5]:
            a, b = rnd(rngs, max(nnz, 1), f32), rnd(rngs, nnz + 2, f32)
            icol = rngs.permutation(nnz + 2)[:max(nnz, 1)].astype(np.int32)
            for alpha, beta in ((f32(1.0), f32(0.0)), (rnd(rngs, 1, f32)[0], rnd(rngs, 1, f32)[0])):
                c0 = rnd(rngs, 1, f32)[0]
                r = f32(L.ktref_csrmm_col_elem_s(ci(bits), ci(nnz), p(a), p(b), p(icol), cf(alpha), cf(beta), cf(c0)))
                cols.append(dict(psz=psz, nnz=nnz, a=hx(a[:nnz]), b=hx(b), icol=[int(c) for c in icol[:nnz]],
                                 alpha=hx(np.array([alpha], f32))[0], beta=hx(np.array([beta], f32))[0],
                                 c0=hx(np.array([c0], f32))[0], out=hx(np.array([r], f32))[0]))
    out["csrmm_col_s"] = cols
    rows_s = []
    for bits, psz in ((256, 8), (512, 16)):
        for n in (1, 7, 8, 9, 15, 16, 17, 35):
            for nnz in (0, 1, 3, 4, 6, 9):
                k = nnz + 2
                a, B = rnd(rngs, max(nnz, 1), f32), rnd(rngs, k * n, f32)
                icol = rngs.permutation(k)[:max(nnz, 1)].astype(np.int32)
                alpha, beta = rnd(rngs, 2, f32)
                c = rnd(rngs, n, f32)
                c0 = c.copy()
                st = L.ktref_csrmm_row_s(ci(bits), ci(nnz), p(a), p(B), ci(n), p(icol), ci(n), cf(alpha), cf(beta), p(c))
                assert st == 0
                rows_s.append(dict(psz=psz, n=n, nnz=nnz, k=k, a=hx(a[:nnz]), B=hx(B), icol=[int(v) for v in icol[:nnz]],
                                   alpha=hx(np.array([alpha], f32))[0], beta=hx(np.array([beta], f32))[0], c0=hx(c0), out=hx(c)))
    out["csrmm_row_s"] = rows_s
    path = os.path.join(HERE, "kt_vectors.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes;", len(hs), "hsum/dot,", len(rows), "trsv rows,", len(col),
          "csrmm col,", len(row), "csrmm row,", len(cols), "float csrmm col,", len(rows_s), "float csrmm row")


if __name__ == "__main__":
    main()
