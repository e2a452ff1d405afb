"""Pin the oracle's restatement of the reference's KT ("kernel template") kernels -- what aoclsparse_?trsv / ?csrmm
dispatch for kid 1/2/3 and by default on an AVX2 / AVX-512 host (trsv.cpp:321-353, csrmm.hpp:779-833) -- against outputs
of the reference's own micro-kernel templates (tests/golden/kt_vectors.json, made by tests/golden/make_kt_vectors.py from
oracle/_ref/libktref.so).  Inputs are random with full mantissas, so a different summation tree gives different bits."""
import ctypes
import json
import os

import numpy as np
import pytest

import oracle

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def ktv():
    with open(os.path.join(HERE, "golden", "kt_vectors.json")) as f:
        return json.load(f)


def unhex(h, dtype=np.float64):
    if isinstance(h, str):
        h = [h]
    if dtype == np.float32:
        return np.array([int(v, 16) for v in h], dtype=np.uint32).view(np.float32)
    return np.array([int(v, 16) for v in h], dtype=np.uint64).view(np.float64)


def bits_equal(a, b):
    a, b = np.atleast_1d(a), np.atleast_1d(b)
    u = np.uint32 if a.dtype == np.float32 else np.uint64
    return a.dtype == b.dtype and np.array_equal(a.view(u), b.view(u))


def test_hsum_and_dot_trees_match_the_reference_templates(ktv):
    # kt_l0_avx2.hpp:331-351, kt_l0_avx512.hpp:367-376 (= gcc's _mm512_reduce_add_*), kt_l1.hpp:41-46
    seen = set()
    for c in ktv["hsum_dot"]:
        dt = np.float64 if c["type"] == "d" else np.float32
        v, w = unhex(c["v"], dt), unhex(c["w"], dt)
        assert bits_equal(np.array([oracle.kt_hsum(c["tsz"], v)], dt), unhex(c["hsum"], dt)), c
        # kt_dot_p = kt_mul_p then kt_hsum_p: one rounding per product, then the same tree
        assert bits_equal(np.array([oracle.kt_hsum(c["tsz"], (v * w).astype(dt))], dt), unhex(c["dot"], dt)), c
        seen.add((c["type"], c["tsz"]))
    assert seen == {("d", 4), ("d", 8), ("s", 8), ("s", 16)}


def _row_system(cnt, a, x, icol, xi, dtype, upper):
    """A triangular system whose LAST-solved row holds the entries (a, icol) and whose other rows are identity rows, so that
    with alpha = 1 and a unit diagonal every x_j it gathers equals b_j exactly and its own right-hand side is xi."""
    m = len(x) + 1
    if not upper:
        ilrow = np.zeros(m + 1, np.int32)
        ilrow[m] = cnt
        idiag = ilrow[:m].copy()
        idiag[m - 1] = cnt
        b = np.concatenate([x, [xi]]).astype(dtype)
        return m, ilrow, idiag, np.asarray(icol, np.int32), b, m - 1
    # U: row 0 is solved last (backward sweep) and depends on rows 1..m-1
    ilrow = np.full(m + 1, cnt, np.int32)
    ilrow[0] = 0
    iurow = ilrow[:m].copy()
    b = np.concatenate([[xi], x]).astype(dtype)
    return m, ilrow, iurow, np.asarray(icol, np.int32) + 1, b, 0


def test_kt_trsv_rows_match_the_reference_sequence(ktv):
    # trsv_kt.cpp:92-137 (kt_trsv_l) and :324-371 (kt_trsv_u): full groups, hsum, masked dot for rem == tsz-1, scalar tail
    kinds = set()
    for c in ktv["trsv_row"]:
        dt = np.float64 if c["type"] == "d" else np.float32
        a, x, xi = unhex(c["a"], dt), unhex(c["x"], dt), unhex(c["xi"], dt)[0]
        for upper in (False, True):
            m, ilrow, ilend, icol, b, at = _row_system(c["cnt"], a, x, c["icol"], xi, dt, upper)
            args = ("u" if upper else "l", c["tsz"], 1.0, m, 0, a if c["cnt"] else np.zeros(1, dt),
                    icol if c["cnt"] else np.zeros(1, np.int32), ilrow, ilend, b, True)
            # the fixture comes from a GCC build with the reference's flags (-march=znver2): scalar tails unfused
            with oracle.contract(False):
                st, sol = oracle.trsv_kt(*args, dtype=dt)
            assert st == 0
            assert bits_equal(sol[at:at + 1], unhex(c["out"], dt)), (c["type"], c["kid"], c["cnt"], upper)
            rem = c["cnt"] % c["tsz"]
            if rem == 0 or rem == c["tsz"] - 1:
                # no scalar tail: the fused (clang / AOCC) build of the reference gives the same bits
                st, sol = oracle.trsv_kt(*args, dtype=dt)
                assert bits_equal(sol[at:at + 1], unhex(c["out"], dt)), (c["type"], c["kid"], c["cnt"], upper, "fused")
        kinds.add((c["type"], c["tsz"], c["cnt"] % c["tsz"] == c["tsz"] - 1, c["cnt"] >= c["tsz"]))
    # every branch of the row sequence was exercised for every vector width
    for t, tsz in (("d", 4), ("d", 8), ("s", 8), ("s", 16)):
        for masked in (False, True):
            assert (t, tsz, masked, True) in kinds and (t, tsz, masked, False) in kinds


def test_kt_trsv_orders_differ_from_the_scalar_chain_and_agree_within_bound():
    # the KT order is a different summation: same result up to the forward-error bound, generally not the same bits
    rng = np.random.default_rng(7)
    m, differ = 400, 0
    from util import triangular_system
    ptr, ind, val = triangular_system(11, m, 14)
    r = oracle.dcsr_optimize(m, m, len(val), 0, ptr, ind, val)
    b = rng.standard_normal(m)
    for kind, ilend in (("l", r["idiag"]), ("u", r["iurow"])):
        st, x0 = oracle.dtrsv(kind, 1.0, m, 0, r["val"], r["ind"], r["ptr"], ilend, b, False)
        for tsz in (4, 8):
            st, x1 = oracle.trsv_kt(kind, tsz, 1.0, m, 0, r["val"], r["ind"], r["ptr"], ilend, b, False)
            assert st == 0
            differ += int(not np.array_equal(x0, x1))
            assert np.allclose(x0, x1, rtol=1e-9, atol=1e-12)
    assert differ > 0
    # transposed KT kernels are the reference kernels' bits (same per-element fma)
    for kind, ilend in (("lt", r["idiag"]), ("ut", r["iurow"])):
        st, x0 = oracle.dtrsv(kind, 1.3, m, 0, r["val"], r["ind"], r["ptr"], ilend, b, False)
        for tsz in (4, 8):
            st, x1 = oracle.trsv_kt(kind, tsz, 1.3, m, 0, r["val"], r["ind"], r["ptr"], ilend, b, False)
            assert np.array_equal(x0, x1)


def test_kt_csrmm_col_elements_match_the_reference_sequence(ktv):
    # csrmm_kt.cpp:127-191
    for c in ktv["csrmm_col"]:
        a, b = unhex(c["a"]), unhex(c["b"])
        alpha, beta, c0 = unhex(c["alpha"])[0], unhex(c["beta"])[0], unhex(c["c0"])[0]
        nnz = c["nnz"]
        row = np.array([0, nnz], np.int32)
        col = np.array(c["icol"] if nnz else [0], np.int32)
        args = ("col", c["psz"], alpha, 0, a if nnz else np.zeros(1), col, row, 1, b, 1, len(b), beta, np.array([c0]), 1)
        with oracle.contract(False):  # GCC -march=znver2 build: the scalar tail is not fused
            st, C = oracle.dcsrmm_kt(*args)
        assert st == 0 and bits_equal(C, unhex(c["out"])), (c["psz"], nnz)
        if nnz % c["psz"] == 0 and alpha == 1.0 and beta == 0.0:
            # no scalar tail and a trivial epilogue: the statement-wise fused reading gives the same bits
            st, C = oracle.dcsrmm_kt(*args)
            assert st == 0 and bits_equal(C, unhex(c["out"])), (c["psz"], nnz, "fused")


def test_kt_csrmm_rows_match_the_reference_sequence(ktv):
    # csrmm_kt.cpp:244-356: vector columns fma(alpha*a, b, c), the n % psz tail columns fma(a*b, alpha, c)
    for c in ktv["csrmm_row"]:
        a, B = unhex(c["a"]), unhex(c["B"])
        alpha, beta = unhex(c["alpha"])[0], unhex(c["beta"])[0]
        n, nnz = c["n"], c["nnz"]
        row = np.array([0, nnz], np.int32)
        col = np.array(c["icol"] if nnz else [0], np.int32)
        st, C = oracle.dcsrmm_kt("row", c["psz"], alpha, 0, a if nnz else np.zeros(1), col, row, 1, B, n, n, beta,
                                 unhex(c["c0"]), n)
        assert st == 0 and bits_equal(C, unhex(c["out"])), (c["psz"], n, nnz)


def test_kt_float_csrmm_matches_the_reference_sequence(ktv):
    # the float instances of the same two templates: 8 lanes (256-bit) and 16 lanes (512-bit)
    f32 = np.float32
    seen = set()
    for c in ktv["csrmm_col_s"]:
        a, b = unhex(c["a"], f32), unhex(c["b"], f32)
        alpha, beta, c0 = unhex(c["alpha"], f32)[0], unhex(c["beta"], f32)[0], unhex(c["c0"], f32)[0]
        nnz = c["nnz"]
        row = np.array([0, nnz], np.int32)
        col = np.array(c["icol"] if nnz else [0], np.int32)
        args = ("col", c["psz"], alpha, 0, a if nnz else np.zeros(1, f32), col, row, 1, b, 1, len(b), beta, np.array([c0], f32), 1)
        with oracle.contract(False):
            st, C = oracle.scsrmm_kt(*args)
        assert st == 0 and bits_equal(C, unhex(c["out"], f32)), (c["psz"], nnz)
        if nnz % c["psz"] == 0 and alpha == 1.0 and beta == 0.0:
            st, C = oracle.scsrmm_kt(*args)
            assert st == 0 and bits_equal(C, unhex(c["out"], f32)), (c["psz"], nnz, "fused")
        seen.add((c["psz"], nnz >= c["psz"], nnz % c["psz"] != 0))
    for psz in (8, 16):
        assert {(psz, True, True), (psz, True, False), (psz, False, True)} <= seen
    for c in ktv["csrmm_row_s"]:
        a, B = unhex(c["a"], f32), unhex(c["B"], f32)
        alpha, beta = unhex(c["alpha"], f32)[0], unhex(c["beta"], f32)[0]
        n, nnz = c["n"], c["nnz"]
        row = np.array([0, nnz], np.int32)
        col = np.array(c["icol"] if nnz else [0], np.int32)
        st, C = oracle.scsrmm_kt("row", c["psz"], alpha, 0, a if nnz else np.zeros(1, f32), col, row, 1, B, n, n, beta,
                                 unhex(c["c0"], f32), n)
        assert st == 0 and bits_equal(C, unhex(c["out"], f32)), (c["psz"], n, nnz)


def test_kt_csrmm_reads_c_when_beta_is_zero():
    # csrmm_kt.cpp:176-191, :246: beta*C is computed even for beta == 0, so NaN / Inf in C propagate
    val, col, row = np.array([2.0]), np.array([0], np.int32), np.array([0, 1], np.int32)
    B = np.array([3.0])
    for order in ("col", "row"):
        for psz in (4, 8):
            st, C = oracle.dcsrmm_kt(order, psz, 1.0, 0, val, col, row, 1, B, 1, 1, 0.0, np.array([np.nan]), 1)
            assert st == 0 and np.isnan(C[0])
            st, C = oracle.dcsrmm_kt(order, psz, 1.0, 0, val, col, row, 1, B, 1, 1, 0.0, np.array([5.0]), 1)
            assert C[0] == 6.0


@pytest.mark.skipif(oracle.ktref() is None, reason="oracle/_ref/libktref.so is built only where /root/reference exists")
def test_live_reference_templates_random_sweep():
    """Where the reference's templates are built (this container), sweep many more random cases than the fixture holds."""
    L = oracle.ktref()
    rng = np.random.default_rng(99)
    P = ctypes.c_void_p
    for _ in range(300):
        for kid, tsz in ((1, 4), (3, 8)):
            cnt = int(rng.integers(0, 40))
            a = rng.standard_normal(max(cnt, 1)) * 2.0 ** rng.integers(-5, 6, max(cnt, 1))
            x = rng.standard_normal(cnt + 2)
            icol = rng.permutation(cnt + 2)[:max(cnt, 1)].astype(np.int32)
            xi = float(rng.standard_normal())
            ref = L.ktref_trsv_row_d(ctypes.c_int(kid), ctypes.c_double(xi), ctypes.c_int(cnt), a.ctypes.data_as(P),
                                     x.ctypes.data_as(P), icol.ctypes.data_as(P))
            m, ilrow, ilend, ic, b, at = _row_system(cnt, a[:cnt], x, icol[:cnt], xi, np.float64, False)
            with oracle.contract(False):
                st, sol = oracle.trsv_kt("l", tsz, 1.0, m, 0, a, ic if cnt else np.zeros(1, np.int32), ilrow, ilend, b, True)
            assert st == 0 and bits_equal(sol[at:at + 1], np.array([ref]))


def test_spmv_orders_pinned_by_inexact_vectors():
    """tests/golden/order_vectors.json: rows whose products are inexact, expected bits from an exact-rational evaluation of the
    reference's instruction sequences (make_order_vectors.py).  Pins the ORDER of rows a2-a5 (scalar chain, 4-lane AVX2 tree,
    8-lane AVX-512 tree, float 8-lane tree) by a vector instead of by reading, for both ways a compiler builds the scalar
    statements (fused: what the GPU kernels reproduce; gcc_znver2: two roundings in the scalar loops)."""
    with open(os.path.join(HERE, "golden", "order_vectors.json")) as f:
        rows = json.load(f)["rows"]
    distinct = 0
    for c in rows:
        dt = np.float64 if c["type"] == "d" else np.float32
        val, x = unhex(c["val"], dt), unhex(c["x"], dt)
        n = c["n"]
        col, row = np.arange(max(n, 1), dtype=np.int32), np.array([0, n], np.int32)
        v = val if n else np.zeros(1, dt)
        xx = x if n else np.zeros(1, dt)
        y0 = np.array([c["y0"]], dt)
        got = {}
        for mode, fused in (("fused", True), ("gcc_znver2", False)):
            with oracle.contract(fused):
                if dt == np.float64:
                    for kname, order in (("ref", "ref"), ("avx2", "lane4"), ("avx512", "lane8")):
                        st, y = oracle.dcsrmv_order(order, 0, c["alpha"], 1, v, col, row, xx, c["beta"], y0)
                        assert st == 0
                        got["%s/%s" % (kname, mode)] = y
                else:
                    for kname, order in (("ref", "ref"), ("avx2", "lane8")):
                        st, y = oracle.scsrmv(order, 0, c["alpha"], 1, v, col, row, xx, c["beta"], y0)
                        assert st == 0
                        got["%s/%s" % (kname, mode)] = y
        for k, h in c["expect"].items():
            assert bits_equal(got[k].astype(dt), unhex(h, dt)), (c["type"], n, k)
        distinct += len(set(c["expect"].values())) > 1
    assert distinct > 20  # the vectors do tell the orders apart
