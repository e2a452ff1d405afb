"""The oracle against the reference's extreme-value known answers (tests/unit_tests/extreme_value_tests.cpp, transcribed as data by
tests/golden/make_fixtures_r5.py): NaN / Inf / max / min through add, sp2m, csrmm (both KT widths and the reference kernel) and the
sparse dot product.  CPU only; the GPU side of the same vectors is tests/test_gpu_r5.py::test_extreme_value_known_answers."""
import json
import os

import numpy as np
import pytest

import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def k5():
    with open(os.path.join(ROOT, "tests", "golden", "reference_kats_r5.json")) as f:
        return json.load(f)


def ev_array(tokens, dtype=np.float64):
    """the fixture's special-value names in the given type ("tmp": max * 8.9885e-24 + 1 evaluated in that type)"""
    fi = np.finfo(dtype)
    special = {"nan": np.nan, "inf": np.inf, "-inf": -np.inf, "max": fi.max, "min": fi.tiny,
               "tmp": dtype(fi.max) * dtype(8.9885e-24) + dtype(1.0)}
    return np.array([special[t] if isinstance(t, str) else t for t in tokens], dtype=dtype)


def ev_match(got, exp, ulps=2):
    """the reference's EXPECT_ARR_MATCH at its one-ulp-scale tolerance: NaN where NaN, the same infinity, else within `ulps`"""
    got, exp = np.asarray(got, dtype=np.float64), np.asarray(exp, dtype=np.float64)
    assert got.shape == exp.shape, (got.shape, exp.shape)
    nan, inf = np.isnan(exp), np.isinf(exp)
    assert np.array_equal(np.isnan(got), nan), (got, exp)
    assert np.array_equal(got[inf], exp[inf]), (got, exp)
    fin = ~nan & ~inf
    assert np.all(np.abs(got[fin] - exp[fin]) <= ulps * np.finfo(np.float64).eps * np.abs(exp[fin])), (got[fin], exp[fin])
    return True


def sorted_csr(ptr, ind, val):
    ind, val = np.array(ind).copy(), np.array(val).copy()
    for i in range(len(ptr) - 1):
        o = np.argsort(ind[ptr[i]:ptr[i + 1]], kind="stable")
        ind[ptr[i]:ptr[i + 1]] = ind[ptr[i]:ptr[i + 1]][o]
        val[ptr[i]:ptr[i + 1]] = val[ptr[i]:ptr[i + 1]][o]
    return ind, val


def test_extreme_value_sp2m_and_add(k5):
    I = k5["init"]
    m = I["m"]
    A = (np.array(I["A_row_ptr"], np.int32), np.array(I["A_col_ind"], np.int32), ev_array(I["A_val"]))
    B = (np.array(I["B_row_ptr"], np.int32), np.array(I["B_col_ind"], np.int32), ev_array(I["B_val"]))
    st, pc, ic, vc = oracle.dcsr2m(m, m, 0, A[0], A[1], A[2], 0, B[0], B[1], B[2])
    assert st == 0
    ic, vc = sorted_csr(pc, ic, vc)  # the reference compares the sorted export (aocl_csr_sorted_export)
    E = k5["sp2m"]
    # values only, as the reference compares them; its structure arrays are one entry short in the last row (fixture note), so
    # the structure is checked against the product's own pattern: rows 0-5 as listed, row 6 = columns {0, 2, 3, 4, 5}
    assert len(vc) == 34 == len(E["C_exp_val"]) and list(pc[:7]) == E["C_exp_row_ptr"][:7] and list(ic[:29]) == E["C_exp_col_ind"][:29]
    assert list(ic[29:]) == [0, 2, 3, 4, 5]
    ev_match(vc, ev_array(E["C_exp_val"]))
    pa, ia, va = oracle.dcsradd((m, m, 0) + A, False, 1.0, (m, m, 0) + B)
    ia, va = sorted_csr(pa, ia, va)
    E = k5["add"]
    assert list(pa) == E["C_exp_row_ptr"] and list(ia) == E["C_exp_col_ind"]
    ev_match(va, ev_array(E["C_exp_val"]))


@pytest.mark.parametrize("kernel", ["ref_row", "kt4", "kt8"])
def test_extreme_value_csrmm(k5, kernel):
    I, E = k5["init"], k5["csrmm"]
    m = I["m"]
    val, col, row = ev_array(I["A_val"]), np.array(I["A_col_ind"], np.int32), np.array(I["A_row_ptr"], np.int32)
    Bd = ev_array(I["B_dense_row_major"])
    C0 = np.zeros(m * m)
    if kernel == "ref_row":
        st, C = oracle.dcsrmm("row", 1.0, 0, val, col, row, m, Bd, m, m, 0.0, C0, m)
    else:
        st, C = oracle.dcsrmm_kt("row", 4 if kernel == "kt4" else 8, 1.0, 0, val, col, row, m, Bd, m, m, 0.0, C0, m)
    assert st == 0
    ev_match(C, ev_array(E["C_exp_val"]))


def test_extreme_value_dot(k5):
    D = k5["dot"]
    indx, y = np.array(D["indx"], np.int32), np.array(D["y"], np.float64)
    for case in D["cases"]:
        x = np.array(D["x"], np.float64)
        x[0], x[1] = ev_array([case["x0"]])[0], ev_array([case["x1"]])[0]
        ev_match(np.array([oracle.ddoti(x, indx, y)]), ev_array([case["expected"]]))


def symm_expected_y(e, diag, x):
    """y = E x for the symmetric matrix the reference's optimize holds (fixture symm_opt: ptr / ind + the values of the diagonal
    type); small integers, so every summation order gives the same doubles"""
    vals = np.array(e[{0: "non_unit_diag_val", 1: "unit_diag_val", 2: "zero_diag_val"}[diag]])
    ptr, ind = e["ptr"], e["ind"]
    return np.array([sum(vals[p] * x[ind[p]] for p in range(ptr[i], ptr[i + 1])) for i in range(len(ptr) - 1)])


@pytest.mark.parametrize("base", [0, 1])
def test_symmetric_mv_matches_the_matrix_the_reference_optimize_builds(k5, base):
    """optimize_symm_herm_tests.cpp:39-938 (real types): for each of its four matrices -- one with unsorted rows and missing
    diagonal entries -- each triangle and each diagonal type, the oracle's symmetric SpMV on the clean CSR equals E x, E being the
    expanded matrix the reference lists as the content of its optimized copy."""
    for M in k5["symm_opt"]["matrices"]:
        m = M["m"]
        rp, ci, v = np.array(M["row_ptr"], np.int32) + base, np.array(M["col_ind"], np.int32) + base, np.array(M["val"])
        o = oracle.dcsr_optimize(m, m, M["nnz"], base, rp, ci, v)
        assert o["status"] == 0
        x = np.arange(1, m + 1, dtype=np.float64) * np.array([1, -2, 3, 5][:m])
        for fill, tri in ((0, "lower"), (1, "upper")):
            for diag in (0, 1, 2):
                st, y = oracle.dcsrmv_special("symm", o["base"], 1.0, m, m, diag, fill, o["val"], o["ind"], o["ptr"], o["idiag"], o["iurow"],
                                              x, 0.0, np.zeros(m))
                assert st == 0 and np.array_equal(y, symm_expected_y(M["expected"][tri], diag, x)), (M["id"], tri, diag)
