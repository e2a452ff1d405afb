"""Round-3 reference vectors (tests/golden/reference_kats_r3.json, provenance in tests/golden/make_fixtures_r3.py) against the
CPU oracle: the remaining csrmm `init` cases, the csr2m gold CSR, the sp2m CSC case and the configurations of the reference's
randomised sp2m tests.  The same fixtures are run through the C ABI on the GPU in tests/test_gpu_parity_r3.py."""
import json
import os

import numpy as np
import pytest

import oracle

HERE = os.path.dirname(os.path.abspath(__file__))
EPS = np.finfo(np.float64).eps


@pytest.fixture(scope="module")
def k3():
    with open(os.path.join(HERE, "golden", "reference_kats_r3.json")) as f:
        return json.load(f)


def ulp_close(got, exp, ulps=4):
    """gtest's EXPECT_DOUBLE_EQ (what csrmm_tests.cpp:2139-2142 uses): within 4 units in the last place"""
    got, exp = np.asarray(got, np.float64), np.asarray(exp, np.float64)
    return np.all(np.abs(got - exp) <= ulps * np.spacing(np.maximum(np.abs(got), np.abs(exp))))


def csr_of(run):
    """the stored arrays as a CSR matrix S: S = A for CSR input, S = A^T (k x m) for CSC input"""
    m, k = run["m"], run["k"]
    ptr, ind, val = np.array(run["ptr"], np.int32), np.array(run["ind"], np.int32), np.array(run["val"], np.float64)
    rows, cols = (k, m) if run["format"] == "csc" else (m, k)
    return ptr, ind, val, rows, cols, run["base"]


def general_product(run, psz=None):
    """op(A) * B for a general matrix through the oracle's csrmm kernels (psz: KT kernel with that many lanes); CSC input
    and op = transpose go through csr2csc as in the reference (csrmm.hpp:737-771)"""
    ptr, ind, val, rows, cols, base = csr_of(run)
    want_transpose_of_S = (run["op"] == "t") != (run["format"] == "csc")
    if want_transpose_of_S:
        st, cp, ri, cv = oracle.dcsr2csc(rows, cols, len(val), base, base, ptr, ind, val)
        assert st == 0
        ptr, ind, val, rows, cols = cp, ri, cv, cols, rows
    n = run["n"]
    order = run["order"]
    if psz is None:
        st, C = oracle.dcsrmm(order, run["alpha"], base, val, ind, ptr, rows, run["B"], n, run["ldb"], run["beta"], run["C"],
                              run["ldc"])
    else:
        st, C = oracle.dcsrmm_kt(order, psz, run["alpha"], base, val, ind, ptr, rows, run["B"], n, run["ldb"], run["beta"],
                                 run["C"], run["ldc"])
    assert st == 0
    return C


def test_csrmm_general_runs(k3):
    # csrmm_tests.cpp:218-291 (id 2), :326-391 (id 4), :392-431 (id 5), :559-590 (id 9, CSC), :614-644 (id 11, CSC)
    seen = set()
    for run in k3["csrmm_runs"]:
        if run["type"] != "general":
            continue
        for psz in (None, 4, 8):     # kid 0, kid 1/2, kid 3: every kernel of the dispatch table must pass the reference's test
            C = general_product(run, psz)
            assert ulp_close(C, run["C_exp"]), (run["id"], run["format"], run["op"], run["order"], psz)
        seen.add(run["id"])
    assert seen == {2, 4, 5, 9, 11}


def test_csrmm_greater_ld(k3):
    # csrmm_tests.cpp:1995-2050: ldb = 2k, ldc = 2m, column-major; the padding rows of C keep their values
    c = k3["csrmm_greater_ld"]
    for psz in (None, 4, 8):
        args = (c["alpha"], 0, c["val"], c["ind"], c["ptr"], c["m"], c["B"], c["n"], c["ldb"], c["beta"], c["C"], c["ldc"])
        st, C = oracle.dcsrmm("col", *args) if psz is None else oracle.dcsrmm_kt("col", psz, *args)
        assert st == 0
        got = np.array(C).reshape(c["n"], c["ldc"])
        exp = np.array(c["C_exp"]).reshape(c["n"], c["ldc"])
        assert ulp_close(got[:, :c["m"]], exp[:, :c["m"]])
        assert np.array_equal(got[:, c["m"]:], np.array(c["C"]).reshape(c["n"], c["ldc"])[:, c["m"]:])


def dense_of(m, n, base, ptr, ind, val):
    A = np.zeros((m, n))
    for i in range(m):
        for p in range(ptr[i] - base, ptr[i + 1] - base):
            A[i, ind[p] - base] += val[p]
    return A


def test_csr2m_gold(k3):
    # csr2m_tests.cpp:216-600: one-based operands in every base mix, C always zero-based; gold CSR incl. column order
    c = k3["csr2m"]
    for ba, bb in c["base_mixes"]:
        pa = np.array(c["A"]["ptr"], np.int32) - 1 + ba
        ia = np.array(c["A"]["ind"], np.int32) - 1 + ba
        pb = np.array(c["B"]["ptr"], np.int32) - 1 + bb
        ib = np.array(c["B"]["ind"], np.int32) - 1 + bb
        st, pc, ic, vc = oracle.dcsr2m(c["m"], c["n"], ba, pa, ia, c["A"]["val"], bb, pb, ib, c["B"]["val"])
        assert st == 0
        assert np.array_equal(pc, c["C"]["ptr"]) and np.array_equal(ic, c["C"]["ind"])
        assert np.array_equal(vc, np.array(c["C"]["val"]))


def test_sp2m_csc_case_and_configurations(k3):
    # sp2m_tests.cpp:371-444: CSC A (= CSR of A^T) times CSR identity
    c = k3["sp2m_csc"]
    a = c["A_csc"]
    st, rp, ci, v = oracle.dcsr2csc(c["n"], c["m"], len(a["val"]), 0, 0, a["ptr"], a["ind"], a["val"])   # CSR of A
    assert st == 0
    b = c["B_csr"]
    st, pc, ic, vc = oracle.dcsr2m(c["m"], c["n"], 0, rp, ci, v, 0, b["ptr"], b["ind"], b["val"])
    assert st == 0
    assert np.array_equal(dense_of(3, 3, 0, pc, ic, vc).ravel(), np.array(c["dense_C"], np.float64))
    # sp2m_tests.cpp:880-1050: the real double configurations (random operands of the stated shape; dense check at sqrt(eps))
    from util import random_csr
    done = 0
    for cfg in k3["sp2m_configs"]["cases"]:
        if cfg["type"] != "d":
            continue
        ma, na, mb, nb = cfg["m_a"], cfg["n_a"], cfg["m_b"], cfg["n_b"]
        rng = np.random.default_rng(cfg["nnz_a"] * 131 + cfg["nnz_b"])

        def rand(m, n, nnz, base):
            cells = rng.choice(m * n, size=min(nnz, m * n), replace=False)
            cells.sort()
            r, cidx = cells // n, cells % n
            ptr = np.zeros(m + 1, np.int32)
            np.add.at(ptr, r + 1, 1)
            return (np.cumsum(ptr).astype(np.int32) + base, (cidx + base).astype(np.int32), rng.uniform(-2, 2, len(cells)))

        pa, ia, va = rand(ma, na, cfg["nnz_a"], cfg["base_a"])
        pb, ib, vb = rand(mb, nb, cfg["nnz_b"], cfg["base_b"])
        A, B = dense_of(ma, na, cfg["base_a"], pa, ia, va), dense_of(mb, nb, cfg["base_b"], pb, ib, vb)

        def operand(m, n, base, ptr, ind, val, op):
            if op == "n":
                return m, n, ptr, ind, val
            st, cp, ri, cv = oracle.dcsr2csc(m, n, len(val), base, base, ptr, ind, val)
            assert st == 0
            return n, m, cp, ri, cv

        m1, k1, p1, i1, v1 = operand(ma, na, cfg["base_a"], pa, ia, va, cfg["op_a"])
        k2, n2, p2, i2, v2 = operand(mb, nb, cfg["base_b"], pb, ib, vb, cfg["op_b"])
        assert k1 == k2, cfg
        st, pc, ic, vc = oracle.dcsr2m(m1, n2, cfg["base_a"], p1, i1, v1, cfg["base_b"], p2, i2, v2)
        assert st == 0
        D = (A if cfg["op_a"] == "n" else A.T) @ (B if cfg["op_b"] == "n" else B.T)
        assert np.allclose(dense_of(m1, n2, 0, pc, ic, vc), D, atol=np.sqrt(EPS), rtol=0), cfg
        done += 1
    assert done >= 4


def test_mv_empty_rows_and_extreme_values_through_the_oracle(k3):
    # mv_tests.cpp:1319-1356
    c = k3["mv_empty_rows"]
    st, y = oracle.dcsrmvt(0, c["alpha"], c["m"], c["n"], c["val"], c["ind"], c["ptr"], c["x"], c["beta"], np.zeros(c["n"]))
    assert st == 0 and np.array_equal(y, c["y_exp"])
    # mv_tests.cpp:1858-2100, general / op = none configurations: NaN and Inf land exactly where the dense expression puts them
    ex = k3["mv_extreme"]
    for cfg in ex["configs"]:
        if cfg["type"] != "general" or cfg["op"] != "n":
            continue
        s = ex["systems"][cfg["system"]]
        val, x = np.array(s["val"], np.float64), np.array(s["x"], np.float64)
        plant(cfg, val, x)
        beta = 0.0 if cfg["beta_zero"] else s["beta"]
        st, y = oracle.dcsrmv(0, 0, s["alpha"], s["n"], len(val), val, s["ind"], s["ptr"], x, beta, s["y0"])
        assert st == 0
        assert classes_match(y, dense_expect(s, cfg, val, x, beta)), cfg


def plant(cfg, val, x):
    special = {"ET_NAN": np.nan, "ET_INF": np.inf, "ET_ZERO": 0.0}
    within = cfg["range"] == "FLOW_EDGE_WITHIN"
    flow = {"ET_POVRFLOW": (1e154, 1e153) if within else (1e200, 1e200),
            "ET_NOVRFLOW": (-1e154, 1e153) if within else (-1e200, 1e200),
            "ET_PUNDRFLOW": (1e-154, 1e-153) if within else (1e-200, 1e-200),
            "ET_NUNDRFLOW": (-1e-154, 1e-153) if within else (-1e-200, 1e-200)}
    if cfg["op1"] in flow:
        val[cfg["val_offset"]], x[cfg["x_offset"]] = flow[cfg["op1"]]
        return
    if cfg["op1"] in special:
        val[cfg["val_offset"]] = special[cfg["op1"]]
    if cfg["op2"] in special:
        x[cfg["x_offset"]] = special[cfg["op2"]]
        if cfg["x2_follows"]:
            x[2] = x[cfg["x_offset"]]


def dense_expect(s, cfg, val, x, beta):
    n = s["n"]
    with np.errstate(all="ignore"):
        y = beta * np.array(s["y0"], np.float64) if beta != 0.0 else np.zeros(n)
        ptr, ind = s["ptr"], s["ind"]
        for i in range(n):
            for p in range(ptr[i], ptr[i + 1]):
                j = ind[p]
                a = val[p]
                if cfg["type"] == "general":
                    r, c = (i, j) if cfg["op"] == "n" else (j, i)
                    y[r] += s["alpha"] * a * x[c]
                    continue
                keep = (j <= i) if cfg["fill"] == "lower" else (j >= i)
                if not keep or (i == j and cfg["diag"] == "zero"):
                    continue
                if cfg["type"] == "triangular":
                    r, c = (i, j) if cfg["op"] == "n" else (j, i)
                    y[r] += s["alpha"] * a * x[c]
                else:  # symmetric: both triangles from the stored one
                    y[i] += s["alpha"] * a * x[j]
                    if i != j:
                        y[j] += s["alpha"] * a * x[i]
    return y


def classes_match(got, exp):
    got, exp = np.asarray(got), np.asarray(exp)
    if not np.array_equal(np.isnan(got), np.isnan(exp)):
        return False
    inf = np.isinf(exp)
    if not np.array_equal(np.isinf(got), inf) or not np.array_equal(np.sign(got[inf]), np.sign(exp[inf])):
        return False
    fin = np.isfinite(exp)
    return np.allclose(got[fin], exp[fin], rtol=1e-12, atol=1e-300)
