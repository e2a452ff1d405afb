"""Pin the CPU oracle (oracle/oracle.c) against the reference's own known-answer vectors
(tests/golden/reference_kats.json; provenance in tests/golden/make_fixtures.py)."""
import numpy as np
import pytest

import oracle
from util import banded_rows, laplace5, random_csr, triangular_system

EPS = np.finfo(np.float64).eps


def test_csrmv_kat(kats):
    # tests/unit_tests/csrmv_tests.cpp:185-220, tests/examples/sample_spmv_c.c:50-59
    for c in kats["csrmv"]:
        nnz = len(c["val"])
        for kid in (-1, 0, 1, 2, 3):
            st, y = oracle.dcsrmv(kid, c["base"], c["alpha"], c["m"], nnz, c["val"], c["col_ind"],
                                  c["row_ptr"], c["x"], c["beta"], c["y0"])
            assert st == 0
            assert np.array_equal(y, np.array(c["y_gold"], dtype=np.float64)), (c["name"], kid)
        for order in ("ref", "lane4", "lane8"):
            st, y = oracle.dcsrmv_order(order, c["base"], c["alpha"], c["m"], c["val"], c["col_ind"],
                                        c["row_ptr"], c["x"], c["beta"], c["y0"])
            assert st == 0 and np.array_equal(y, np.array(c["y_gold"], dtype=np.float64))


def test_csrmv_beta0_ignores_nan_in_y(kats):
    # csrmv_kr.hpp:504-509: y is not read when beta == 0 (mv_tests.cpp:1858-2290)
    c = kats["csrmv"][0]
    y0 = np.full(c["m"], np.nan)
    st, y = oracle.dcsrmv(-1, 0, 1.0, c["m"], 8, c["val"], c["col_ind"], c["row_ptr"], c["x"], 0.0, y0)
    assert st == 0 and np.array_equal(y, np.array(c["y_gold"], dtype=np.float64))


def test_clean_csr_kat(kats):
    # tests/unit_tests/hint_tests.cpp:75-170: clean CSR must be integer-exact
    for c in kats["clean_csr"]:
        nnz = len(c["val"])
        r = oracle.dcsr_optimize(c["m"], c["n"], nnz, 0, c["row_ptr"], c["col_ind"], c["val"])
        e = c["exp"]
        assert r["status"] == 0, c["name"]
        assert r["is_internal"] == e["is_internal"], c["name"]
        assert np.array_equal(r["ptr"], e["icrow"]), c["name"]
        assert np.array_equal(r["ind"], e["icol"]), c["name"]
        assert np.array_equal(r["val"], np.array(e["aval"], dtype=np.float64)), c["name"]
        dim = min(c["m"], c["n"])
        assert np.array_equal(r["idiag"][:dim], e["idiag"]), c["name"]
        assert np.array_equal(r["iurow"][:dim], e["iurow"]), c["name"]


def _solve(kats_case, base):
    c = kats_case
    m = c["m"]
    ptr = np.array(c["row_ptr"], dtype=np.int32) + base
    ind = np.array(c["col_ind"], dtype=np.int32) + base
    val = np.array(c["val"], dtype=np.float64)
    r = oracle.dcsr_optimize(m, m, len(val), base, ptr, ind, val)
    assert r["status"] == 0
    kind = {("lower", "n"): "l", ("lower", "t"): "lt", ("upper", "n"): "u", ("upper", "t"): "ut"}[
        (c["fill"], c["trans"])]
    ilend = r["idiag"] if c["fill"] == "lower" else r["iurow"]
    st, x = oracle.dtrsv(kind, c["alpha"], m, r["base"], r["val"], r["ind"], r["ptr"], ilend,
                         c["b"], c["unit"])
    assert st == 0
    return x


def test_trsv_kat(kats):
    # tests/unit_tests/common_data_utils.h:1373-2305 via trsv_tests.cpp:279-318
    tol = kats["trsv_abs_tol"]
    assert len(kats["trsv"]) == 24
    for c in kats["trsv"]:
        for base in (0, 1):
            x = _solve(c, base)
            err = np.max(np.abs(x - np.array(c["xref"])))
            assert err <= tol, (c["name"], base, err)


def test_trsv_strided(kats):
    c = [t for t in kats["trsv"] if t["name"] == "S7_Lx_aB"][0]
    m = c["m"]
    r = oracle.dcsr_optimize(m, m, len(c["val"]), 0, c["row_ptr"], c["col_ind"], c["val"])
    incb, incx = 3, 2
    b = np.zeros(m * incb)
    b[::incb] = c["b"]
    st, x = oracle.dtrsv("l", c["alpha"], m, 0, r["val"], r["ind"], r["ptr"], r["idiag"], b,
                         c["unit"], incb=incb, incx=incx)
    assert st == 0
    assert np.max(np.abs(x[::incx][:m] - np.array(c["xref"]))) <= kats["trsv_abs_tol"]


def test_csrmm_kat(kats):
    # tests/unit_tests/csrmm_tests.cpp:99-325
    for c in kats["csrmm"]:
        m, k, n = c["m"], c["k"], c["n"]
        for order, ldb, ldc, key in (("col", k, m, "C_exp_col"), ("row", n, n, "C_exp_row")):
            if c["alpha"] == 0.0:
                st, C = oracle.dscale_dense(order, c["C"], m, n, ldc, c["beta"])
            else:
                st, C = oracle.dcsrmm(order, c["alpha"], 0, c["val"], c["col_ind"], c["row_ptr"], m,
                                      c["B"], n, ldb, c["beta"], c["C"], ldc)
            assert st == 0
            exp = np.array(c[key], dtype=np.float64)
            assert np.allclose(C[: m * n], exp, rtol=1e-13, atol=1e-12), (c["name"], order)


def test_csrmm_transpose_kat_via_csr2csc(kats):
    # op = transpose KATs (csrmm_tests.cpp:190-215): the reference transposes A with csr2csc and
    # runs the same kernels (csrmm.hpp:737-771).
    c = [t for t in kats["csrmm"] if t["name"] == "id1_5x5"][0]
    m, k, n = c["m"], c["k"], c["n"]
    st, cp, ri, cv = oracle.dcsr2csc(m, k, len(c["val"]), 0, 0, c["row_ptr"], c["col_ind"], c["val"])
    assert st == 0
    for order, ld, key in (("col", 5, "C_exp_col_T"), ("row", 5, "C_exp_row_T")):
        st, C = oracle.dcsrmm(order, c["alpha"], 0, cv, ri, cp, k, c["B"], n, ld, c["beta"], c["C"], ld)
        assert st == 0
        assert np.allclose(C, np.array(c[key]), rtol=1e-13, atol=1e-12), order


def _rand_csr(rng, m, n, row_nnz, base=0):
    ptr = [0]
    ind, val = [], []
    for i in range(m):
        k = int(row_nnz(i))
        cols = np.sort(rng.choice(n, size=min(k, n), replace=False))
        ind += list(cols)
        val += list(rng.uniform(-1, 1, size=len(cols)))
        ptr.append(len(ind))
    return (np.array(ptr, np.int32) + base, np.array(ind, np.int32) + base,
            np.array(val, np.float64))


def test_csrmv_orders_within_forward_error_bound():
    """All three reference orders agree within c*eps*sum|a_ij x_j|."""
    rng = np.random.default_rng(69069)
    m = n = 600
    ptr, ind, val = _rand_csr(rng, m, n, lambda i: rng.integers(0, 90))
    x = rng.uniform(-1, 1, n)
    y0 = rng.uniform(-1, 1, m)
    ref = None
    absrow = np.zeros(m)
    for i in range(m):
        s, e = ptr[i], ptr[i + 1]
        absrow[i] = np.sum(np.abs(val[s:e] * x[ind[s:e]]))
    exact = np.array([np.sum(val[ptr[i]:ptr[i + 1]].astype(np.longdouble)
                             * x[ind[ptr[i]:ptr[i + 1]]].astype(np.longdouble)) for i in range(m)])
    for order in ("ref", "lane4", "lane8"):
        st, y = oracle.dcsrmv_order(order, 0, 1.0, m, val, ind, ptr, x, 0.0, y0)
        assert st == 0
        rowlen = np.diff(ptr)
        bound = (rowlen + 2) * EPS * absrow + 1e-300
        assert np.all(np.abs(y - exact.astype(np.float64)) <= bound), order
        ref = y if ref is None else ref
    # base-1 gives bitwise the same result as base-0
    st, y1 = oracle.dcsrmv_order("lane8", 1, 1.0, m, val, ind + 1, ptr + 1, x, 0.0, y0)
    st, y0_ = oracle.dcsrmv_order("lane8", 0, 1.0, m, val, ind, ptr, x, 0.0, y0)
    assert np.array_equal(y1, y0_)


def test_dispatch_rule_and_kid():
    # csrmv.hpp:332-333: nnz <= 10*m forces the scalar kernel whatever kid says
    rng = np.random.default_rng(1)
    m = n = 200
    ptr, ind, val = _rand_csr(rng, m, n, lambda i: 9)
    x = rng.uniform(-1, 1, n)
    st, ya = oracle.dcsrmv(3, 0, 1.0, m, len(val), val, ind, ptr, x, 0.0, np.zeros(m))
    st, yr = oracle.dcsrmv_order("ref", 0, 1.0, m, val, ind, ptr, x, 0.0, np.zeros(m))
    assert np.array_equal(ya, yr)
    ptr, ind, val = _rand_csr(rng, m, n, lambda i: 40)
    st, ya = oracle.dcsrmv(-1, 0, 1.0, m, len(val), val, ind, ptr, x, 0.0, np.zeros(m))
    st, y8 = oracle.dcsrmv_order("lane8", 0, 1.0, m, val, ind, ptr, x, 0.0, np.zeros(m))
    assert np.array_equal(ya, y8)
    st, _ = oracle.dcsrmv(4, 0, 1.0, m, len(val), val, ind, ptr, x, 0.0, np.zeros(m))
    assert st == 14  # aoclsparse_status_invalid_kid
    # OpenMP leg computes the same bits
    st, yo = oracle.dcsrmv(-1, 0, 2.5, m, len(val), val, ind, ptr, x, 0.5, np.ones(m), nthreads=2)
    st, ys = oracle.dcsrmv(-1, 0, 2.5, m, len(val), val, ind, ptr, x, 0.5, np.ones(m))
    assert np.array_equal(yo, ys)


def test_transposed_spmv_matches_dense():
    rng = np.random.default_rng(5)
    m, n = 70, 50
    ptr, ind, val = _rand_csr(rng, m, n, lambda i: rng.integers(0, 20))
    A = np.zeros((m, n))
    for i in range(m):
        A[i, ind[ptr[i]:ptr[i + 1]]] = val[ptr[i]:ptr[i + 1]]
    x = rng.uniform(-1, 1, m)
    y0 = rng.uniform(-1, 1, n)
    st, y = oracle.dcsrmvt(0, 5.1, m, n, val, ind, ptr, x, 3.2, y0)
    assert st == 0 and np.allclose(y, 5.1 * A.T @ x + 3.2 * y0, rtol=1e-13, atol=1e-13)


def test_mat_check_classes():
    # csr_util.cpp:124-279
    st, sort, fd = oracle.mat_check(5, 5, 8, [0, 2, 3, 4, 7, 8], [0, 3, 1, 2, 1, 3, 4, 4], [1.0] * 8, 0, 0)
    assert (st, sort, fd) == (0, 1, True)
    st, sort, fd = oracle.mat_check(5, 5, 8, [0, 2, 3, 4, 7, 8], [3, 0, 1, 2, 3, 1, 4, 4], [1.0] * 8, 0, 0)
    assert (st, sort, fd) == (0, 3, True)
    st, sort, fd = oracle.mat_check(5, 5, 9, [0, 2, 3, 4, 8, 9], [0, 3, 1, 2, 2, 1, 3, 4, 4], [1.0] * 9, 0, 0)
    assert (st, sort, fd) == (0, 2, True)
    # out-of-range column -> invalid_index_value; duplicate diagonal -> invalid_value
    assert oracle.mat_check(2, 2, 2, [0, 1, 2], [0, 2], [1.0, 1.0], 0, 0)[0] == 6
    assert oracle.mat_check(2, 2, 3, [0, 2, 3], [0, 0, 1], [1.0] * 3, 0, 0)[0] == 5
    assert oracle.mat_check(2, 2, 2, [1, 1, 2], [0, 1], [1.0] * 2, 0, 0)[0] == 5  # ptr[0] != base


def test_ilu0_and_trsv_roundtrip():
    """ILU(0) of a 5-pt Laplacian: L*U reproduces A on A's pattern; L solve has tiny residual."""
    g = 12
    m = g * g
    rows, cols, vals = [], [], []
    ptr = [0]
    for r in range(m):
        i, j = divmod(r, g)
        for c, v in ((r - g, -1.0), (r - 1, -1.0), (r, 4.0), (r + 1, -1.0), (r + g, -1.0)):
            if c < 0 or c >= m or (c == r - 1 and j == 0) or (c == r + 1 and j == g - 1):
                continue
            cols.append(c)
            vals.append(v)
        ptr.append(len(cols))
    st, lu, diag = oracle.dilu0(m, 0, ptr, cols, vals)
    assert st == 0
    L = np.eye(m)
    U = np.zeros((m, m))
    A = np.zeros((m, m))
    for r in range(m):
        for p in range(ptr[r], ptr[r + 1]):
            A[r, cols[p]] = vals[p]
            if cols[p] < r:
                L[r, cols[p]] = lu[p]
            else:
                U[r, cols[p]] = lu[p]
    P = (A != 0)
    assert np.allclose((L @ U)[P], A[P], atol=1e-12)
    r = oracle.dcsr_optimize(m, m, len(cols), 0, ptr, cols, lu)
    assert not r["is_internal"]
    b = L @ np.ones(m)
    st, x = oracle.dtrsv("l", 1.0, m, 0, lu, cols, ptr, r["idiag"], b, True)
    assert st == 0 and np.max(np.abs(x - 1.0)) < 1e-13


def test_sp2m_matches_dense():
    rng = np.random.default_rng(11)
    m, k, n = 40, 30, 50
    pa, ia, va = _rand_csr(rng, m, k, lambda i: rng.integers(0, 8))
    pb, ib, vb = _rand_csr(rng, k, n, lambda i: rng.integers(0, 8), base=1)
    st, pc, ic, vc = oracle.dcsr2m(m, n, 0, pa, ia, va, 1, pb, ib, vb)
    assert st == 0
    A = np.zeros((m, k))
    B = np.zeros((k, n))
    for i in range(m):
        A[i, ia[pa[i]:pa[i + 1]]] = va[pa[i]:pa[i + 1]]
    for i in range(k):
        B[i, ib[pb[i] - 1:pb[i + 1] - 1] - 1] = vb[pb[i] - 1:pb[i + 1] - 1]
    C = np.zeros((m, n))
    for i in range(m):
        assert len(set(ic[pc[i]:pc[i + 1]])) == pc[i + 1] - pc[i]
        C[i, ic[pc[i]:pc[i + 1]]] = vc[pc[i]:pc[i + 1]]
    assert np.allclose(C, A @ B, atol=1e-13)


def _symgs_oracle(c, fill, trans, base=0, x0=None, iters=None):
    m = c["m"]
    rp = np.array(c["row_ptr"], np.int32) + base
    ci = np.array(c["col_ind"], np.int32) + base
    v = np.array(c["val"], np.float64)
    o = oracle.dcsr_optimize(m, m, len(v), base, rp, ci, v)
    assert o["status"] == 0 and o["fulldiag"]
    x = np.array(c["x0"] if x0 is None else x0, np.float64)
    mtype = {"general": 0, "symmetric": 1, "triangular": 3}[c["mtype"]]
    for _ in range(c["iters"] if iters is None else iters):
        st, x = oracle.dsymgs(mtype, fill, trans, o["base"], c["alpha"], m, o["val"], o["ind"], o["ptr"], o["idiag"],
                              o["iurow"], np.array(c["b"], np.float64), x)
        assert st == 0
    return x


def test_symgs_kat(kats):
    """symgs_tests.cpp:380-455 on the systems of common_data_utils.h:2673-3895: the sweep and the closing
    product, every (fill, op) the reference runs, within the reference's own tolerance
    expected_precision(10) = 10*sqrt(2 eps) and, tighter, 1e-13 relative (the vectors carry 17 digits)."""
    tol = kats["trsv_abs_tol"]
    for c in kats["symgs"]:
        m = c["m"]
        rp, ci, v = np.array(c["row_ptr"], np.int32), np.array(c["col_ind"], np.int32), np.array(c["val"], np.float64)
        for fill in (0, 1):
            for trans in (0, 1):
                x = _symgs_oracle(c, fill, trans, base=fill)  # both index bases along the way
                xg = c["x_gold"] if c["mtype"] == "symmetric" else c["x_gold"]["nt"[trans]]
                yg = c["y_gold"] if c["mtype"] == "symmetric" else c["y_gold"]["nt"[trans]]
                xg, yg = np.array(xg, np.float64), np.array(yg, np.float64)
                assert np.all(np.abs(x - xg) <= tol), (c["name"], fill, trans)
                assert np.all(np.abs(x - xg) <= 1e-13 * np.maximum(1.0, np.abs(xg)).max() * 4), (c["name"], fill, trans)
                if trans == 0 or c["mtype"] == "symmetric":
                    so, y = oracle.dcsrmv(0, 0, 1.0, m, len(v), v, ci, rp, x, 0.0, np.zeros(m))
                else:
                    so, y = oracle.dcsrmvt(0, 1.0, m, m, v, ci, rp, x, 0.0, np.zeros(m))
                assert np.all(np.abs(y - yg) <= tol * max(1.0, np.abs(yg).max())), (c["name"], fill, trans)


def test_ilu_solve_inverts_the_factors():
    m, rp, ci, v = laplace5(12)
    st, lu, diag = oracle.dilu0(m, 0, rp, ci, v)
    assert st == 0
    b = np.random.default_rng(5).uniform(-1, 1, m)
    st, x = oracle.dilu_solve(m, 0, diag, lu, rp, ci, b)
    assert st == 0
    Lm, Um = np.eye(m), np.zeros((m, m))
    for i in range(m):
        for p in range(rp[i], rp[i + 1]):
            if ci[p] < i:
                Lm[i, ci[p]] = lu[p]
            else:
                Um[i, ci[p]] = lu[p]
    assert np.allclose(Lm @ (Um @ x), b, rtol=0, atol=1e-12)


def test_ell_kat_and_conversions(kats):
    """ellmv_tests.cpp:151-252: csr2ell of the one-based 3x3 gives the -1 padded arrays the test also feeds
    directly, and both give y_gold; ELLT and ELLT-HYB of the same matrix agree."""
    c = kats["ell"][0]
    m, base = c["m"], c["base"]
    w, ec, ev = oracle.csr2ell("ell", m, base, c["row_ptr"], c["col_ind"], c["val"])
    assert w == c["ell_width"] and list(ec) == c["ell_col_ind"] and list(ev) == c["ell_val"]
    st, y = oracle.dellmv("ell", base, c["alpha"], m, ev, ec, w, c["x"], c["beta"], np.full(m, np.nan))
    assert st == 0 and list(y) == c["y_gold"]
    st, yf = oracle.sellmv(base, c["alpha"], m, ev, ec, w, c["x"], c["beta"], np.full(m, np.nan))
    assert st == 0 and list(yf) == c["y_gold"]
    w, tc, tv = oracle.csr2ell("ellt", m, base, c["row_ptr"], c["col_ind"], c["val"])
    st, y = oracle.dellmv("ellt", base, c["alpha"], m, tv, tc, w, c["x"], c["beta"], np.full(m, np.nan))
    assert st == 0 and list(y) == c["y_gold"]
    w, em, mp, hc, hv = oracle.csr2ell("hyb", m, base, c["row_ptr"], c["col_ind"], c["val"])
    assert (w, em, list(mp)) == (1, 2, [2])
    st, y = oracle.dellthybmv(base, c["alpha"], m, hv, hc, w, em, c["val"], c["row_ptr"], c["col_ind"], mp, c["x"],
                              c["beta"], np.full(m, np.nan))
    assert st == 0 and list(y) == c["y_gold"]


def test_ell_orders_against_csr_oracle():
    rng = np.random.default_rng(31)
    m, n = 300, 280
    lens = rng.integers(0, 23, m)
    lens[5] = 0
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    ci = np.concatenate([np.sort(rng.choice(n, k, replace=False)) for k in lens]).astype(np.int32)
    v = rng.uniform(-1, 1, len(ci))
    x, y0 = rng.uniform(-1, 1, n), rng.uniform(-1, 1, m)
    # row-major ELL in the 4-lane order == CSR 4-lane kernel on rows whose length is a multiple of 4 or < 4 ...
    w, ec, ev = oracle.csr2ell("ell", m, 0, rp, ci, v)
    st, ye = oracle.dellmv("ell", 0, 1.7, m, ev, ec, w, x, -0.3, y0)
    st, yc = oracle.dcsrmv_order("lane4", 0, 1.7, m, v, ci, rp, x, -0.3, y0)
    assert np.array_equal(ye, yc)  # ... and in fact on every row: both walk full groups then a scalar tail
    w, tc, tv = oracle.csr2ell("ellt", m, 0, rp, ci, v)
    st, yt = oracle.dellmv("ellt", 0, 1.7, m, tv, tc, w, x, -0.3, y0)
    st, yr = oracle.dcsrmv_order("ref", 0, 1.7, m, v, ci, rp, x, -0.3, y0)
    assert np.allclose(yt, yr, rtol=0, atol=1e-13)  # padding adds +0*x terms: same value unless -0/Inf
    w, em, mp, hc, hv = oracle.csr2ell("hyb", m, 0, rp, ci, v)
    assert em + len(mp) == m and np.all(lens[mp] > w) and np.count_nonzero(lens <= w) == em
    st, yh = oracle.dellthybmv(0, 1.7, m, hv, hc, w, em, v, rp, ci, mp, x, -0.3, y0)
    assert np.array_equal(yh[mp], yc[mp]) and np.allclose(yh, yr, rtol=0, atol=1e-13)


def test_csrsv_restatement_equals_the_pinned_trsv_chain():
    """aoclsparse_csrsv.hpp:88-187 has no vectors in the reference's tests; on sorted rows with a full diagonal its loops
    are the chains of ref_trsv_l / ref_trsv_u, which ARE pinned (trsv KATs): the two restatements must agree bitwise."""
    m = 400
    rp, ci, v = triangular_system(71, m, 5)
    b = np.random.default_rng(3).uniform(-1, 1, m)
    o = oracle.dcsr_optimize(m, m, len(v), 0, rp, ci, v)
    assert o["status"] == 0
    for lower in (True, False):
        for unit in (False, True):
            st, y = oracle.dcsrsv(lower, unit, 1.7, m, v, ci, rp, b)
            st2, x = oracle.dtrsv("l" if lower else "u", 1.7, m, o["base"], o["val"], o["ind"], o["ptr"],
                                  o["idiag"] if lower else o["iurow"], b, unit)
            assert st == 0 and st2 == 0 and np.array_equal(y, x[:m])


def test_blkcsr_kats(kats):
    """blkcsrmv_tests.cpp:444-470 (block arrays fed directly) and :518-537 / :656-676 (CSR through csr2blkcsr
    for 1/2/4 x 8 blocks): every route gives y_gold, and the 2x8 conversion of the one-based CSR reproduces the
    direct arrays of the first test."""
    d = kats["blkcsr"]["direct"]
    st, y = oracle.dblkcsrmv(d["base"], d["alpha"], d["m"], d["masks"], d["val"], d["blk_col_ind"], d["blk_row_ptr"],
                             d["x"], d["beta"], np.full(d["m"], np.nan), d["rows_blk"])
    assert st == 0 and list(y) == d["y_gold"]
    for c in kats["blkcsr"]["csr"]:
        for rows in (1, 2, 4):
            st, brp, bc, bv, mk = oracle.csr2blkcsr(c["m"], c["n"], c["base"], c["row_ptr"], c["col_ind"], c["val"], rows)
            assert st == 0
            if rows == 2 and c["base"] == 1:
                assert list(brp) == d["blk_row_ptr"] and list(bc) == d["blk_col_ind"] and list(mk) == d["masks"]
                assert list(bv) == d["val"]
            st, y = oracle.dblkcsrmv(c["base"], c["alpha"], c["m"], mk, bv, bc, brp, c["x"], c["beta"],
                                     np.full(c["m"], np.nan), rows)
            assert st == 0 and list(y) == c["y_gold"]
    # error returns pinned by blkcsrmv_tests.cpp:776-900 (opt_blksize gives 0, csr2blkcsr the size code)
    assert oracle.opt_blksize(0, 3, 0, [0, 1, 1], [1])[0] == 0 and oracle.opt_blksize(2, -1, 0, [0, 1, 1], [1])[0] == 0
    assert oracle.csr2blkcsr(2, 7, 0, [0, 1, 1], [1], [3.0], 2)[0] == 3
    assert oracle.csr2blkcsr(2, 8, 0, [0, 1, 1], [1], [3.0], 3)[0] == 3


def test_blkcsr_structure_properties():
    """conversion invariants on random banded matrices: every entry lands in exactly one block bit, windows stay
    inside [0, n), popcounts add up to nnz, and the product equals the CSR product up to the 8-lane regrouping."""
    for seed, base, n in ((1, 0, 64), (2, 1, 21), (3, 0, 8)):
        m = 37
        rp, ci, v = banded_rows(seed, m, n, lambda r, i: 0 if i % 9 == 4 else 3 + (i * 7) % 11, base)
        x, y0 = np.random.default_rng(5).uniform(-1, 1, n), np.random.default_rng(6).uniform(-1, 1, m)
        st, yr = oracle.dcsrmv_order("ref", base, 1.3, m, v, ci, rp, x, -0.4, y0)
        for rows in (1, 2, 4):
            st, brp, bc, bv, mk = oracle.csr2blkcsr(m, n, base, rp, ci, v, rows)
            assert st == 0 and int(np.unpackbits(mk).sum()) == len(v)
            assert np.all(bc - base >= 0) and np.all(bc - base + 8 <= n)
            # rebuild (row, col, val) triples from the blocks and compare with the CSR
            trip, iv = [], 0
            for i0 in range(0, m, rows):
                for b in range(brp[i0] - base, brp[i0 + 1] - base):
                    for r in range(rows):
                        for l in range(8):
                            if mk[b * rows + r] >> l & 1:
                                trip.append((i0 + r, bc[b] - base + l, bv[iv]))
                                iv += 1
            ref = [(i, ci[p] - base, v[p]) for i in range(m) for p in range(rp[i] - base, rp[i + 1] - base)]
            assert sorted(trip) == sorted(ref)
            st, y = oracle.dblkcsrmv(base, 1.3, m, mk, bv, bc, brp, x, -0.4, y0, rows)
            assert st == 0 and np.allclose(y, yr, rtol=0, atol=1e-13)
        r, tot = oracle.opt_blksize(m, len(v), base, rp, ci)
        if r:
            assert tot == len(oracle.csr2blkcsr(m, n, base, rp, ci, v, r)[2])


def _sym_full(n, rp, ci, v):
    """lower-triangle CSR -> full symmetric CSR (sorted rows)"""
    rows = [[] for _ in range(n)]
    for i in range(n):
        for p in range(rp[i], rp[i + 1]):
            rows[i].append((ci[p], v[p]))
            if ci[p] != i:
                rows[ci[p]].append((i, v[p]))
    rows = [sorted(r) for r in rows]
    frp = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int32)
    return frp, np.array([c for r in rows for c, _ in r], np.int32), np.array([a for r in rows for _, a in r], np.float64)


def test_itsol_examples(kats):
    """sample_itsol_d_cg.cpp: CG + SGS on the 8x8 SPD system reaches `expected` within its 5e-6 absolute tolerance;
    restarted GMRES (+ILU0) on cage4 reaches the 0.5 vector.  Pins the restated solvers on the reference's examples."""
    c = kats["itsol"]["cg"]
    n = c["n"]
    rp, ci, v = _sym_full(n, c["row_ptr"], c["col_ind"], c["val"])
    o = oracle.dcsr_optimize(n, n, len(v), 0, rp, ci, v)
    xe = np.array(c["expected"])
    so, b = oracle.dcsrmv(0, 0, 1.0, n, len(v), v, ci, rp, xe, 0.0, np.zeros(n))
    for precond in (0, 3):
        st, x, rinfo = oracle.dcg(n, 0, o["ptr"], o["ind"], o["val"], o["idiag"], o["iurow"], b, c["x0"], 4.2e-8,
                                  c["abs_tol"], 500, precond)
        assert st == 0 and rinfo[0] <= c["abs_tol"] and np.max(np.abs(x - xe)) < 1e-5 and 1 <= rinfo[30] <= 8
    g = kats["itsol"]["gmres"]
    n = g["n"]
    rp, ci, v = np.array(g["row_ptr"], np.int32), np.array(g["col_ind"], np.int32), np.array(g["val"])
    xe = np.full(n, 0.5)
    so, b = oracle.dcsrmv(0, 0, 1.0, n, len(v), v, ci, rp, xe, 0.0, np.zeros(n))
    for precond in (0, 2):
        st, x, rinfo = oracle.dgmres(n, 0, rp, ci, v, b, np.ones(n), 7, 4.2e-8, 1e-10, 50, precond)
        assert st == 0 and np.max(np.abs(x - xe)) < 1e-6, (precond, st, rinfo[0], rinfo[30])


def test_complex_mv_restatement_on_a_hand_checked_hermitian_matrix():
    """3x3 hermitian H = [[2, 1-i, 0], [1+i, 3, 2i], [0, -2i, 1]] stored as its lower triangle: H x, H^T x = conj(H) x
    and H^H x = H x worked out by hand."""
    ptr, ind = [0, 1, 3, 5], [0, 0, 1, 1, 2]
    val = [2, 1 + 1j, 3, -2j, 1]
    x = np.array([1, 1j, 2])
    y, _ = oracle.zmv("n", "hermitian", "lower", "non_unit", 0, 1.0, 3, 3, ptr, ind, val, x, 0.0, np.zeros(3))
    assert np.allclose(y, [2 + (1 - 1j) * 1j, (1 + 1j) + 3j + 4j, -2j * 1j + 2])
    yt, _ = oracle.zmv("t", "hermitian", "lower", "non_unit", 0, 1.0, 3, 3, ptr, ind, val, x, 0.0, np.zeros(3))
    assert np.allclose(yt, [2 + (1 + 1j) * 1j, (1 - 1j) + 3j - 4j, 2j * 1j + 2])
    yh, _ = oracle.zmv("h", "hermitian", "lower", "non_unit", 0, 1.0, 3, 3, ptr, ind, val, x, 0.0, np.zeros(3))
    assert np.allclose(yh, y)


# --------------------------------------------------------------------------------------------------
# dense-result product, CSR -> dense, sparse sum (sp2md.hpp, convert.hpp:658-929, csradd.hpp)
# --------------------------------------------------------------------------------------------------
def _scipy_csr(m, n, base, ptr, ind, val):
    import scipy.sparse as sp
    return sp.csr_matrix((np.array(val, dtype=np.float64), np.asarray(ind) - base, np.asarray(ptr) - base), shape=(m, n))  # copies: scipy sorts in place


def test_spmmd_and_csr2dense_kats(kats):
    """The reference's own vectors: spmmd_tests.cpp:130-157 and conversion_tests.cpp:211-252."""
    k = kats["spmmd"]
    for base_a in (0, 1):
        for base_b in (0, 1):
            a = (3, 3, base_a, np.array(k["a"]["row_ptr"]) + base_a, np.array(k["a"]["col_ind"]) + base_a, k["a"]["val"])
            b = (3, 3, base_b, np.array(k["b"]["row_ptr"]) + base_b, np.array(k["b"]["col_ind"]) + base_b, k["b"]["val"])
            for trans, gold in ((False, k["c_none"]), (True, k["c_trans"])):
                G = np.array(gold, dtype=np.float64).reshape(3, 3)
                c = oracle.dsp2md(a, trans, b, False, 1.0, 0.0, np.full(9, np.nan), True, 3)
                assert np.array_equal(c.reshape(3, 3), G)
                c = oracle.dsp2md(a, trans, b, False, 1.0, 0.0, np.full(12, 7.0), False, 4)  # column-major, ldc 4
                assert np.array_equal(c.reshape(3, 4)[:, :3].T, G) and np.all(c.reshape(3, 4)[:, 3] == 7.0)
    k = kats["csr2dense"]
    args = (5, 5, 0, k["row_ptr"], k["col_ind"], k["val"])
    assert np.array_equal(oracle.dcsr2dense(*args, np.full(25, -1.0), 5, False), k["rowmajor"])
    assert np.array_equal(oracle.dcsr2dense(*args, np.full(25, -1.0), 5, True), k["colmajor"])
    assert np.array_equal(oracle.dcsr2dense(*args, np.zeros(40), 8, False), k["rowmajor_ld8"])
    one = (5, 5, 1, np.array(k["row_ptr"]) + 1, np.array(k["col_ind"]) + 1, k["val"])
    assert np.array_equal(oracle.dcsr2dense(*one, np.full(25, -1.0), 5, False), k["rowmajor"])


def test_dense_result_restatements_against_scipy():
    """Independent cross-check of the three restatements on random unsorted operands (scipy is not the reference:
    it pins the mathematics, the KATs above pin the layout conventions)."""
    m, k, n = 60, 45, 50
    for base_a, base_b in ((0, 0), (1, 0), (0, 1)):
        pa, ia, va = random_csr(11, m, k, lambda r, i: r.integers(0, 9), base=base_a, sort=False)
        pb, ib, vb = random_csr(12, k, n, lambda r, i: r.integers(0, 9), base=base_b, sort=False)
        pt, it, vt = random_csr(13, m, n, lambda r, i: r.integers(0, 9), base=base_b, sort=False)
        A, B, Bt = _scipy_csr(m, k, base_a, pa, ia, va), _scipy_csr(k, n, base_b, pb, ib, vb), _scipy_csr(m, n, base_b, pt, it, vt)
        c0 = np.random.default_rng(5).uniform(-1, 1, m * n)
        c = oracle.dsp2md((m, k, base_a, pa, ia, va), False, (k, n, base_b, pb, ib, vb), False, 1.5, -0.5, c0, True, n)
        assert np.allclose(c.reshape(m, n), 1.5 * (A @ B).toarray() - 0.5 * c0.reshape(m, n), atol=1e-13)
        c0 = np.random.default_rng(6).uniform(-1, 1, k * n)
        c = oracle.dsp2md((m, k, base_a, pa, ia, va), True, (m, n, base_b, pt, it, vt), False, 2.0, 1.0, c0, False, k)
        assert np.allclose(c.reshape(n, k).T, 2.0 * (A.T @ Bt).toarray() + c0.reshape(n, k).T, atol=1e-13)
        c = oracle.dsp2md((m, k, base_a, pa, ia, va), False, (n, k, base_b, *random_csr(14, n, k, lambda r, i: 4, base=base_b)),
                          True, 1.0, 0.0, np.zeros(m * n), True, n)
        B3 = _scipy_csr(n, k, base_b, *random_csr(14, n, k, lambda r, i: 4, base=base_b))
        assert np.allclose(c.reshape(m, n), (A @ B3.T).toarray(), atol=1e-13)
        # sparse sum: structure = A's row then B's new columns in order; values exact against scipy's sum
        pb2, ib2, vb2 = random_csr(15, m, k, lambda r, i: r.integers(0, 9), base=base_b, sort=False)
        pc, ic, vc = oracle.dcsradd((m, k, base_a, pa, ia, va), False, 0.75, (m, k, base_b, pb2, ib2, vb2))
        S = 0.75 * A + _scipy_csr(m, k, base_b, pb2, ib2, vb2)
        assert pc[0] == base_a and np.allclose(_scipy_csr(m, k, base_a, pc, ic, vc).toarray(), S.toarray(), atol=1e-15)
        for i in range(m):
            row = ic[pc[i] - base_a:pc[i + 1] - base_a]
            la = pa[i + 1] - pa[i]
            assert np.array_equal(row[:la], ia[pa[i] - base_a:pa[i + 1] - base_a]) and len(set(row)) == len(row)
        pc, ic, vc = oracle.dcsradd((k, m, base_a, *random_csr(16, k, m, lambda r, i: 5, base=base_a)), True, -2.0,
                                    (m, k, base_b, pb2, ib2, vb2))
        At = _scipy_csr(k, m, base_a, *random_csr(16, k, m, lambda r, i: 5, base=base_a))
        assert np.allclose(_scipy_csr(m, k, base_a, pc, ic, vc).toarray(),
                           (-2.0 * At.T + _scipy_csr(m, k, base_b, pb2, ib2, vb2)).toarray(), atol=1e-15)
        # csr2dense variants: symmetric / triangular from one triangle, unit / zero diagonal
        ps, is_, vs = random_csr(17, m, m, lambda r, i: r.integers(1, 9), base=base_a, sort=False)
        D = _scipy_csr(m, m, base_a, ps, is_, vs).toarray()
        lo, up = np.tril(D, -1), np.triu(D, 1)
        got = oracle.dcsr2dense(m, m, base_a, ps, is_, vs, np.zeros(m * m), m, False, 1, 0, 0).reshape(m, m)
        assert np.array_equal(got, lo + lo.T + np.diag(np.diag(D)))
        got = oracle.dcsr2dense(m, m, base_a, ps, is_, vs, np.zeros(m * m), m, True, 1, 1, 1).reshape(m, m)
        assert np.array_equal(got, up + up.T + np.eye(m))
        got = oracle.dcsr2dense(m, m, base_a, ps, is_, vs, np.zeros(m * m), m, False, 3, 1, 2).reshape(m, m)
        assert np.array_equal(got, up)


def test_level1_kats(kats):
    """axpyi_tests.cpp:78-88, roti_tests.cpp:50-99, dotp_tests.cpp:64-69, gthr_tests.cpp:66-71, sctr_tests.cpp:71-84."""
    k = kats["level1"]
    a = k["axpyi"]
    for nnz, gold in ((4, a["y_nnz4"]), (2, a["y_nnz2"])):
        st, y = oracle.daxpyi(a["a"], a["x"][:nnz], a["indx"][:nnz], a["y"])
        assert st == 0 and np.array_equal(y, gold)
    st, y = oracle.daxpyi(2.0, [1.0, 1.0, 1.0], [1, -1, 2], np.zeros(4))
    assert st == 6 and np.array_equal(y, [0, 2, 0, 0])  # entries before the bad index are applied
    for r in k["roti"]:
        st, x, y = oracle.droti(r["x"], r["indx"], r["y"], r["c"], r["s"])
        assert st == 0 and np.array_equal(x, r["x_exp"]) and np.array_equal(y, r["y_exp"])
    d = k["doti"]
    assert abs(oracle.ddoti(d["x"], d["indx"], d["y"]) - d["dot"]) <= 64 * EPS * 271.5
    g = k["gthr"]
    x, y = oracle.gthr(np.array(g["y"], np.float64), g["indx"])
    assert np.array_equal(x, g["x_exp"]) and np.array_equal(y, g["y"])
    x, y = oracle.gthr(np.array(g["y"], np.float64), g["indx"], zero=True)
    assert np.array_equal(x, g["x_exp"]) and np.array_equal(y, g["y_gthrz"])
    s = k["sctr"]
    assert np.array_equal(oracle.sctr(s["x"], s["indx"], np.zeros(17)), s["y_nnz17"])
    assert np.array_equal(oracle.sctr(s["x"], s["indx"][:10], np.zeros(17)), s["y_nnz10"])


def test_dia_bsr_kats_and_properties(kats):
    """diamv_tests.cpp:137-197 / bsrmv_tests.cpp:40-101, and the two formats against the pinned CSR SpMV oracle on random
    matrices (same matrix, different storage: the products agree to a few ulp of sum|a x|)."""
    k = kats["dia_bsr"]
    for base in (0, 1):
        rp, ci = np.array(k["row_ptr"]) + base, np.array(k["col_ind"]) + base
        nd, off, dv = oracle.csr2dia(5, 5, base, rp, ci, k["val"])
        assert nd == 3 and list(off) == [-3, -1, 0]
        assert np.array_equal(oracle.ddiamv(1.0, 5, 5, dv, off, k["x"][:5], 0.0, np.full(5, np.nan)), k["y_gold"][:5])
        for rowmajor in (False, True):
            bp, bi, bv = oracle.csr2bsr(5, 5, base, rp, ci, k["val"], 2, rowmajor)
            assert list(bp - base) == [0, 1, 3, 4] and list(bi - base) == [0, 0, 1, 1]
            if not rowmajor:
                assert np.array_equal(oracle.dbsrmv(1.0, 3, 2, base, bv, bi, bp, k["x"], 0.0, np.full(6, np.nan)), k["y_gold"])
        blk = oracle.csr2bsr(5, 5, base, rp, ci, k["val"], 2, True)[2].reshape(-1, 2, 2)
        assert np.array_equal(blk[0], [[6, 0], [0, 1]]) and np.array_equal(blk[1], [[0, 2], [5, 0]])
    m, n = 301, 257
    for base in (0, 1):
        rp, ci, v = random_csr(91, m, n, lambda r, i: r.integers(0, 9), base=base, sort=False)
        x, y0 = np.random.default_rng(1).uniform(-1, 1, n + 8), np.random.default_rng(2).uniform(-1, 1, m + 8)
        so, yr = oracle.dcsrmv(0, base, 1.5, m, len(v), v, ci, rp, x[:n], -0.5, y0[:m])
        scale = np.abs(_scipy_csr(m, n, base, rp, ci, v)) @ np.abs(x[:n]) * 1.5 + 0.5 * np.abs(y0[:m])
        nd, off, dv = oracle.csr2dia(m, n, base, rp, ci, v)
        assert nd == len(np.unique((ci - base) - np.repeat(np.arange(m), np.diff(rp)))) and np.all(np.diff(off) > 0)
        assert np.all(np.abs(oracle.ddiamv(1.5, m, n, dv, off, x[:n], -0.5, y0[:m]) - yr) <= 16 * EPS * scale + 1e-300)
        for dim in (2, 3, 7, 16):
            mb, nb = (m + dim - 1) // dim, (n + dim - 1) // dim
            bp, bi, bv = oracle.csr2bsr(m, n, base, rp, ci, v, dim, False)
            assert all(np.all(np.diff(bi[bp[r] - base:bp[r + 1] - base]) > 0) for r in range(mb))
            xx, yy = np.zeros(nb * dim), np.zeros(mb * dim)
            xx[:n], yy[:m] = x[:n], y0[:m]
            got = oracle.dbsrmv(1.5, mb, dim, base, bv, bi, bp, xx, -0.5, yy)
            assert np.all(np.abs(got[:m] - yr) <= 16 * EPS * scale + 1e-300) and np.all(got[m:] == 0)
            assert np.count_nonzero(bv) == len(v)


def test_sorv_kats(kats):
    """sorv_tests.cpp:366-414 (1 and 10 sweeps, alpha = 0) and sample_dsorv.cpp:51-61, tolerance of the reference's own
    check (10 * sqrt(2 eps) relative)."""
    tol = 10 * np.sqrt(2 * EPS)
    for k in kats["sorv"]:
        x = np.array(k["x0"], np.float64)
        st, x = oracle.dsorv(k["n"], 0, k["row_ptr"], k["col_ind"], k["val"], k["omega"], 1.0, x, k["b"])
        assert st == 0 and np.allclose(x, k["x_iter1"], rtol=tol, atol=tol)
        if "x_iter10" in k:
            for _ in range(9):
                st, x = oracle.dsorv(k["n"], 0, k["row_ptr"], k["col_ind"], k["val"], k["omega"], 1.0, x, k["b"])
            assert np.allclose(x, k["x_iter10"], rtol=tol, atol=tol)
            st, x = oracle.dsorv(k["n"], 0, k["row_ptr"], k["col_ind"], k["val"], k["omega"], 0.0, x, k["b"])
            assert np.allclose(x, k["x_iter10_then_alpha0"], rtol=tol, atol=tol)
    st, _ = oracle.dsorv(2, 0, [0, 1, 2], [0, 0], [1.0, 1.0], 1.0, 1.0, np.zeros(2), np.zeros(2))
    assert st == 5  # second row has no diagonal


def test_mv_triangular_rectangular_kat(kats):
    """mv_tests.cpp:344-388 (test_mv_success): aoclsparse_?mv with a triangular descriptor on a 5 x 4 matrix, lower and
    upper fill; the oracle's triangular product on the clean CSR must return the reference's exp_y_l / exp_y_u."""
    k = kats["mv_tri"]
    m, n = k["m"], k["n"]
    rp, ci, v = np.array(k["row_ptr"], np.int32), np.array(k["col_ind"], np.int32), np.array(k["val"], np.float64)
    o = oracle.dcsr_optimize(m, n, len(v), k["base"], rp, ci, v)
    assert o["status"] == 0
    x = np.array(k["x"], np.float64)
    for fill, gold in ((0, k["exp_y_l"]), (1, k["exp_y_u"])):
        st, y = oracle.dcsrmv_special("tri", k["base"], k["alpha"], m, n, 0, fill, o["val"], o["ind"], o["ptr"], o["idiag"],
                                      o["iurow"], x, k["beta"], np.full(m, np.nan))
        assert st == 0 and np.array_equal(y, np.array(gold, np.float64))
