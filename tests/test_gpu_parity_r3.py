"""GPU tier, round 3: the reference vectors of tests/golden/reference_kats_r3.json through the C ABI (the way the reference's own
tests call: hint + optimize, then the executor), and the HIP TRSV / csrmm results against the oracle's restatement of the KT
kernels the reference dispatches on AVX2 / AVX-512 hosts (tests/test_oracle_kt.py pins that restatement on the reference's
templates).  Tolerances are written where they are used."""
import ctypes
import json
import os

import numpy as np
import pytest

import oracle
from test_oracle_golden_r3 import classes_match, dense_expect, dense_of, plant, ulp_close
from util import EPS64, kt_lanes, pkg, random_csr, triangular_system

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
P = pkg()
L = P.lib()
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def k3():
    with open(os.path.join(HERE, "golden", "reference_kats_r3.json")) as f:
        return json.load(f)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


OPS = {"n": P.OP_NONE, "t": P.OP_TRANSPOSE, "h": P.OP_CONJ_TRANSPOSE}
TYPES = {"general": P.TYPE_GENERAL, "symmetric": P.TYPE_SYMMETRIC, "triangular": P.TYPE_TRIANGULAR}
FILLS = {"lower": P.FILL_LOWER, "upper": P.FILL_UPPER}
DIAGS = {"non_unit": P.DIAG_NON_UNIT, "unit": P.DIAG_UNIT, "zero": P.DIAG_ZERO}


class Handle:
    """CSR or CSC handle over arrays kept alive here"""

    def __init__(self, fmt, base, m, n, ptr, ind, val):
        self.ptr, self.ind = np.ascontiguousarray(ptr, np.int32), np.ascontiguousarray(ind, np.int32)
        self.val = np.ascontiguousarray(val, np.float64)
        self.h = ctypes.c_void_p()
        fn = L.aoclsparse_create_dcsr if fmt == "csr" else L.aoclsparse_create_dcsc
        self.status = fn(ctypes.byref(self.h), base, m, n, len(self.val), P._ptr(self.ptr), P._ptr(self.ind), P._ptr(self.val))

    def __del__(self):
        try:
            if self.h:
                L.aoclsparse_destroy(ctypes.byref(self.h))
        except Exception:
            pass


def test_csrmm_reference_runs_ids_2_4_5_7_to_11(k3):
    """csrmm_tests.cpp:2055-2170 run_csrmm_case: descriptor, CSR or CSC handle, memory hint, mm hint, optimize, csrmm with the
    stated kid, EXPECT_DOUBLE_EQ_VEC (4 ulp) -- for every (id, format, type, fill, diag, op, order) of the fixture, both
    memory policies, host and device operands, op = transpose and conjugate-transpose."""
    ids = set()
    for run in k3["csrmm_runs"]:
        for mem in (0, 1):
            for op in ([run["op"]] if run["op"] == "n" else ["t", "h"]):
                A = Handle(run["format"], run["base"], run["m"], run["k"], run["ptr"], run["ind"], run["val"])
                assert A.status == 0, run
                d = P.Descr(base=run["base"], mtype=TYPES[run["type"]], fill=FILLS[run["fill"]], diag=DIAGS[run["diag"]])
                assert L.aoclsparse_set_memory_hint(A.h, mem) == 0
                assert L.aoclsparse_set_mm_hint(A.h, OPS[op], d.h, 1000) == 0
                assert L.aoclsparse_optimize(A.h) == 0
                order = P.ORDER_ROW if run["order"] == "row" else P.ORDER_COLUMN
                B, C = np.array(run["B"], np.float64), np.array(run["C"], np.float64)
                for on_device in (False, True):
                    Bx, Cx = (dev(B), dev(C)) if on_device else (B, C.copy())
                    st = P.dcsrmm(OPS[op], run["alpha"], A, d, order, Bx, run["n"], run["ldb"], run["beta"], Cx, run["ldc"],
                                  kid=run["kid"])
                    assert st == 0, (run["id"], run["format"], run["type"], op, run["order"], P.STATUS.get(st, st))
                    got = Cx.cpu().numpy() if on_device else Cx
                    assert ulp_close(got, run["C_exp"]), (run["id"], run["format"], run["type"], run["fill"], run["diag"], op,
                                                          run["order"], on_device)
        ids.add(run["id"])
    assert ids == {2, 4, 5, 7, 8, 9, 10, 11}


def test_csrmm_greater_ld_reference_case(k3):
    # csrmm_tests.cpp:1995-2050 (kid 1 / 3 there): padded ldb = 2k, ldc = 2m; the padding keeps its values
    c = k3["csrmm_greater_ld"]
    A = P.Matrix(0, c["m"], c["k"], c["ptr"], c["ind"], np.array(c["val"], np.float64))
    d = P.Descr()
    for kid in (0, 1, 3):
        C = np.array(c["C"], np.float64)
        assert P.dcsrmm(P.OP_NONE, c["alpha"], A, d, P.ORDER_COLUMN, np.array(c["B"], np.float64), c["n"], c["ldb"], c["beta"],
                        C, c["ldc"], kid=kid) == 0
        assert ulp_close(C, c["C_exp"]), kid


def _export(h):
    base, m, n, nnz = ctypes.c_int(), ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
    rp, ci, v = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
    st = L.aoclsparse_export_dcsr(h, ctypes.byref(base), ctypes.byref(m), ctypes.byref(n), ctypes.byref(nnz), ctypes.byref(rp),
                                  ctypes.byref(ci), ctypes.byref(v))
    assert st == 0
    k = max(nnz.value, 1)
    row = np.ctypeslib.as_array(ctypes.cast(rp, ctypes.POINTER(ctypes.c_int32)), (m.value + 1,)).copy()
    col = np.ctypeslib.as_array(ctypes.cast(ci, ctypes.POINTER(ctypes.c_int32)), (k,))[: nnz.value].copy()
    val = np.ctypeslib.as_array(ctypes.cast(v, ctypes.POINTER(ctypes.c_double)), (k,))[: nnz.value].copy()
    return base.value, m.value, n.value, nnz.value, row, col, val


def test_csr2m_reference_gold(k3):
    """csr2m_tests.cpp:216-600: the gold CSR (zero-based, column order included) for one- and two-stage calls and the three
    base mixes; an invalid base in either descriptor -> invalid_value (:470-491)."""
    c = k3["csr2m"]
    for ba, bb in c["base_mixes"]:
        A = P.Matrix(ba, c["m"], c["k"], np.array(c["A"]["ptr"]) - 1 + ba, np.array(c["A"]["ind"]) - 1 + ba,
                     np.array(c["A"]["val"], np.float64))
        B = P.Matrix(bb, c["k"], c["n"], np.array(c["B"]["ptr"]) - 1 + bb, np.array(c["B"]["ind"]) - 1 + bb,
                     np.array(c["B"]["val"], np.float64))
        dA, dB = P.Descr(base=ba), P.Descr(base=bb)
        for stage in c["stages"]:
            C = ctypes.c_void_p()
            if stage == "full":
                assert L.aoclsparse_dcsr2m(P.OP_NONE, dA.h, A.h, P.OP_NONE, dB.h, B.h, P.STAGE_FULL, ctypes.byref(C)) == 0
            else:
                assert L.aoclsparse_dcsr2m(P.OP_NONE, dA.h, A.h, P.OP_NONE, dB.h, B.h, P.STAGE_NNZ_COUNT, ctypes.byref(C)) == 0
                assert L.aoclsparse_dcsr2m(P.OP_NONE, dA.h, A.h, P.OP_NONE, dB.h, B.h, P.STAGE_FINALIZE, ctypes.byref(C)) == 0
            b, cm, cn, cz, row, col, val = _export(C)
            assert (b, cm, cn, cz) == (0, c["m"], c["n"], len(c["C"]["val"]))
            assert np.array_equal(row, c["C"]["ptr"]) and np.array_equal(col, c["C"]["ind"])
            assert np.array_equal(val, np.array(c["C"]["val"]))
            assert L.aoclsparse_destroy(ctypes.byref(C)) == 0


def test_sp2m_reference_csc_case_and_configurations(k3):
    """sp2m_tests.cpp:371-444 (CSC x CSR, hand-computed dense result) and the real-type configurations of :880-1050 on random
    operands of the stated shape, checked as the reference does: dense op(A) op(B) within sqrt(eps) (:501, :560-585) -- and,
    tighter, bit for bit against the oracle where both operands are CSR with op = none."""
    c = k3["sp2m_csc"]
    A = Handle("csc", 0, c["m"], c["n"], c["A_csc"]["ptr"], c["A_csc"]["ind"], c["A_csc"]["val"])
    B = Handle("csr", 0, c["m"], c["n"], c["B_csr"]["ptr"], c["B_csr"]["ind"], c["B_csr"]["val"])
    d = P.Descr()
    C = ctypes.c_void_p()
    st = L.aoclsparse_sp2m(P.OP_NONE, d.h, A.h, P.OP_NONE, d.h, B.h, P.STAGE_FULL, ctypes.byref(C))
    assert st == 0
    _, cm, cn, _, row, col, val = _export(C)
    assert np.array_equal(dense_of(cm, cn, 0, row, col, val).ravel(), np.array(c["dense_C"], np.float64))
    L.aoclsparse_destroy(ctypes.byref(C))
    done = 0
    for cfg in k3["sp2m_configs"]["cases"]:
        if cfg["type"] != "d":
            continue
        rng = np.random.default_rng(cfg["nnz_a"] * 131 + cfg["nnz_b"])

        def rand(m, n, nnz, base):
            cells = np.sort(rng.choice(m * n, size=min(nnz, m * n), replace=False))
            r, cidx = cells // n, cells % n
            ptr = np.zeros(m + 1, np.int64)
            np.add.at(ptr, r + 1, 1)
            return (np.cumsum(ptr) + base).astype(np.int32), (cidx + base).astype(np.int32), rng.uniform(-2, 2, len(cells))

        ma, na, mb, nb = cfg["m_a"], cfg["n_a"], cfg["m_b"], cfg["n_b"]
        pa, ia, va = rand(ma, na, cfg["nnz_a"], cfg["base_a"])
        pb, ib, vb = rand(mb, nb, cfg["nnz_b"], cfg["base_b"])
        DA, DB = dense_of(ma, na, cfg["base_a"], pa, ia, va), dense_of(mb, nb, cfg["base_b"], pb, ib, vb)

        def handle(csr, base, m, n, ptr, ind, val):
            if csr:
                return Handle("csr", base, m, n, ptr, ind, val)
            st, cp, ri, cv = oracle.dcsr2csc(m, n, len(val), base, base, ptr, ind, val)   # the same matrix in CSC
            assert st == 0
            return Handle("csc", base, m, n, cp, ri, cv)

        A = handle(cfg.get("csr_a", True), cfg["base_a"], ma, na, pa, ia, va)
        B = handle(cfg.get("csr_b", True), cfg["base_b"], mb, nb, pb, ib, vb)
        assert A.status == 0 and B.status == 0
        dA, dB = P.Descr(base=cfg["base_a"]), P.Descr(base=cfg["base_b"])
        C = ctypes.c_void_p()
        if cfg.get("stage", "two-stage") == "full":
            assert L.aoclsparse_sp2m(OPS[cfg["op_a"]], dA.h, A.h, OPS[cfg["op_b"]], dB.h, B.h, P.STAGE_FULL, ctypes.byref(C)) == 0
        else:
            assert L.aoclsparse_sp2m(OPS[cfg["op_a"]], dA.h, A.h, OPS[cfg["op_b"]], dB.h, B.h, P.STAGE_NNZ_COUNT,
                                     ctypes.byref(C)) == 0
            assert L.aoclsparse_sp2m(OPS[cfg["op_a"]], dA.h, A.h, OPS[cfg["op_b"]], dB.h, B.h, P.STAGE_FINALIZE,
                                     ctypes.byref(C)) == 0
        b, cm, cn, _, row, col, val = _export(C)
        D = (DA if cfg["op_a"] == "n" else DA.T) @ (DB if cfg["op_b"] == "n" else DB.T)
        assert b == 0 and (cm, cn) == D.shape, cfg
        assert np.allclose(dense_of(cm, cn, 0, row, col, val), D, atol=np.sqrt(EPS64), rtol=0), cfg
        if cfg.get("csr_a", True) and cfg.get("csr_b", True) and cfg["op_a"] == "n" and cfg["op_b"] == "n":
            so, pc, ic, vc = oracle.dcsr2m(ma, nb, cfg["base_a"], pa, ia, va, cfg["base_b"], pb, ib, vb)
            assert so == 0 and np.array_equal(row, pc) and np.array_equal(col, ic) and np.array_equal(val, vc)
        L.aoclsparse_destroy(ctypes.byref(C))
        done += 1
    assert done >= 4


def test_mv_empty_rows_after_optimize_reference_case(k3):
    # mv_tests.cpp:1319-1356: mv hint for op = none, then the product with op = transpose on a matrix of (almost) empty rows
    c = k3["mv_empty_rows"]
    for dt in (np.float64, np.float32):
        A = P.Matrix(0, c["m"], c["n"], c["ptr"], c["ind"], np.array(c["val"], dt))
        d = P.Descr()
        assert L.aoclsparse_set_mv_hint(A.h, P.OP_NONE, d.h, 1) == 0 and L.aoclsparse_optimize(A.h) == 0
        y = np.zeros(c["n"], dt)
        fn = P.dmv if dt == np.float64 else P.smv
        assert fn(P.OP_TRANSPOSE, c["alpha"], A, d, np.array(c["x"], dt), c["beta"], y) == 0
        assert np.array_equal(y, np.array(c["y_exp"], dt))


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("wmax", [1, 3, 5, 8])
def test_sell_short_row_kernel_large_launch(wmax, dtype):
    """sell_mv_short_kernel (round 3: widest slice <= 8 cells, >= 4,096 slices, scalar order): a random matrix of 300,000 rows
    with <= wmax entries each -- plain SELL-64 (no two rows share a list), slices narrower than the widest one, a band of 300
    consecutive empty rows (slices of width 0), a partial last slice -- and a stencil of the same size (shared lists, modes 0 / 1 /
    2 of the slice words).  Bit for bit against the scalar chain, alpha / beta classes, NaN in x reaching exactly its rows."""
    import __graft_entry__ as entry
    m = 300000 + 37
    # (mostly full-width rows, so that the SELL padding stays inside the budget of optimize)
    rp, ci, v = random_csr(70 + wmax, m, m,
                           lambda r, i: 0 if 1000 <= i < 1300 else (wmax if r.random() < 0.85 else r.integers(0, wmax + 1)))
    mats = [("random", m, rp, ci, v)]
    if wmax == 5:
        ml, rpl, cil, vl = entry.laplace5(550)
        mats.append(("stencil", ml, rpl, cil, vl))
    rng = np.random.default_rng(3)
    for name, mm, rp, ci, v in mats:
        v = v.astype(dtype)
        A = P.Matrix(0, mm, mm, rp, ci, v)
        d = P.Descr()
        assert L.aoclsparse_set_mv_hint(A.h, P.OP_NONE, d.h, 100) == 0 and L.aoclsparse_optimize(A.h) == 0
        assert A.spmv_info().kernel in (3, 4), name
        x = rng.uniform(-1, 1, mm).astype(dtype)
        x[12345] = np.nan
        y0 = rng.uniform(-1, 1, mm).astype(dtype)
        for alpha, beta in ((1.0, 0.0), (-0.5, 1.25)):
            y = y0.copy()
            fn = P.dmv if dtype == np.float64 else P.smv
            assert fn(P.OP_NONE, alpha, A, d, x, beta, y) == 0
            if dtype == np.float64:
                st, yr = oracle.dcsrmv(0, 0, alpha, mm, len(v), v, ci, rp, x, beta, y0.copy())
            else:  # the float kernel of the reference is the 8-lane one: the scalar chain for rows of < 8 entries
                st, yr = oracle.scsrmv("lane8", 0, alpha, mm, v, ci, rp, x, beta, y0.copy())
            assert st == 0
            both_nan = np.isnan(y) & np.isnan(yr)
            assert np.array_equal(np.isnan(y), np.isnan(yr)), (name, wmax)
            assert np.array_equal(y[~both_nan], yr[~both_nan]), (name, wmax, alpha, beta)


def test_mv_extreme_values_reference_configurations(k3):
    """mv_tests.cpp:1858-2100 (real types): NaN * x, Inf * x, Inf * 0 and the overflow / underflow products planted in the
    5 x 5 systems of common_data_utils.h:3897-4075, :4236-4330 for general / symmetric / triangular descriptors x fill x op x
    beta in {2, 0}: NaN and +-Inf must appear exactly where the plain expression puts them (EXPECT_ARR_MATCH), finite entries
    within 1e-12 relative."""
    ex = k3["mv_extreme"]
    n_checked = 0
    for cfg in ex["configs"]:
        s = ex["systems"][cfg["system"]]
        val, x = np.array(s["val"], np.float64), np.array(s["x"], np.float64)
        plant(cfg, val, x)
        beta = 0.0 if cfg["beta_zero"] else s["beta"]
        exp = dense_expect(s, cfg, val, x, beta)
        A = P.Matrix(0, s["n"], s["n"], s["ptr"], s["ind"], val)
        d = P.Descr(mtype=TYPES[cfg["type"]], fill=FILLS[cfg["fill"]], diag=DIAGS[cfg["diag"]])
        for on_device in (False, True):
            y = np.array(s["y0"], np.float64)
            if on_device:
                yd = dev(y)
                st = P.dmv(OPS[cfg["op"]], s["alpha"], A, d, dev(x), beta, yd)
                torch.cuda.synchronize()
                y = yd.cpu().numpy()
            else:
                st = P.dmv(OPS[cfg["op"]], s["alpha"], A, d, x, beta, y)
            assert st == 0, cfg
            assert classes_match(y, exp), (cfg, y, exp)
        n_checked += 1
    assert n_checked >= 35


# ---------------------------------------------------------------------------------------------------------------------------
# KT orders: what the reference's dispatcher runs on an AVX2 / AVX-512 host
# ---------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("fill,op", [("lower", "n"), ("upper", "n"), ("lower", "t"), ("upper", "t")])
def test_trsv_within_bound_of_the_kt_orders(fill, op):
    """aoclsparse_dtrsv (every kid: all of them keep the kid-0 chain) against kt_trsv_{l,u,lt,ut} restated in
    the oracle for 256- and 512-bit vectors (trsv_kt.cpp:64-531), in both builds of the reference (fused / GCC-znver2 scalar
    tails).  Tolerance: |x_gpu - x_kt| <= 8 (maxlen + 4) eps |T^-1| (|alpha b| + |T_off| |x|) componentwise, evaluated with
    the dense comparison solve; the transposed solves are the same per-element fma in KT and reference kernels: bit-identical."""
    m = 600
    rp, ci, v = triangular_system(31, m, 11)
    A = P.Matrix(0, m, m, rp, ci, v)
    r = oracle.dcsr_optimize(m, m, len(v), 0, rp, ci, v)
    b = np.random.default_rng(5).uniform(-1, 1, m)
    alpha = 1.25
    d = P.Descr(mtype=P.TYPE_TRIANGULAR, fill=FILLS[fill])
    kind = {("lower", "n"): "l", ("upper", "n"): "u", ("lower", "t"): "lt", ("upper", "t"): "ut"}[(fill, op)]
    ilend = r["idiag"] if fill == "lower" else r["iurow"]
    D = dense_of(m, m, 0, r["ptr"], r["ind"], r["val"])
    T = np.tril(D) if fill == "lower" else np.triu(D)
    if op == "t":
        T = T.T
    # comparison matrix M(T) = |diag| - |off|: M^-1 >= |T^-1| entrywise for these diagonally dominant systems
    Mc = 2 * np.diag(np.abs(np.diag(T))) - np.abs(T)
    maxlen = int(np.max(np.count_nonzero(T, axis=1)))
    for kid in (-1, 0, 1, 2, 3):
        x = np.zeros(m)
        assert P.dtrsv(OPS[op], alpha, A, d, b, x, kid=None if kid < 0 else kid) == 0
        for tsz in (4, 8):
            for fused in (True, False):
                with oracle.contract(fused):
                    st, xk = oracle.trsv_kt(kind, tsz, alpha, m, 0, r["val"], r["ind"], r["ptr"], ilend, b, False)
                assert st == 0
                if op == "t" and fused:
                    assert np.array_equal(x, xk), (kid, tsz)
                    continue
                rhs = np.abs(alpha * b) + (np.abs(T) - np.diag(np.abs(np.diag(T)))) @ np.abs(xk) + np.abs(np.diag(T) * xk)
                bound = 8 * (maxlen + 4) * EPS64 * np.linalg.solve(Mc, rhs)
                assert np.all(np.abs(x - xk) <= bound), (kid, tsz, fused, float(np.max(np.abs(x - xk) / bound)))


@pytest.mark.parametrize("n", [5, 8, 33])
@pytest.mark.parametrize("alpha,beta", [(1.0, 0.0), (-2.5, 0.75)])
def test_csrmm_within_bound_of_the_kt_orders(n, alpha, beta):
    """aoclsparse_dcsrmm_kid (kid 0..3, both layouts) against csrmm_col_kt / csrmm_row_kt restated for 4 and 8 lanes
    (csrmm_kt.cpp:31-363), both builds: kid 1 / 2 / 3 are BIT-IDENTICAL to the KT kernel of their vector width (fused build);
    every kid is within the tolerance of every order.  Tolerance per element: (len + 4) eps |alpha| sum |a_ik b_kj| + 3 eps |beta c_ij|
    (two summation orders of the same len products + the scaling / beta operations)."""
    m, k = 500, 420
    rp, ci, v = random_csr(77, m, k, lambda r, i: r.integers(0, 40))
    A = P.Matrix(0, m, k, rp, ci, v)
    d = P.Descr()
    rng = np.random.default_rng(12)
    lens = np.diff(rp)
    absA = dense_of(m, k, 0, rp, ci, np.abs(v))
    for order, oname in ((P.ORDER_COLUMN, "col"), (P.ORDER_ROW, "row")):
        ldb, ldc = (k, m) if oname == "col" else (n, n)
        B = rng.uniform(-1, 1, k * n)
        C0 = rng.uniform(-1, 1, m * n)
        Bm = B.reshape(n, k).T if oname == "col" else B.reshape(k, n)
        Cm = C0.reshape(n, m).T if oname == "col" else C0.reshape(m, n)
        bound = ((lens[:, None] + 4) * EPS64 * abs(alpha) * (absA @ np.abs(Bm)) + 3 * EPS64 * np.abs(beta * Cm))
        for kid in (0, 1, 2, 3):
            C = C0.copy()
            assert P.dcsrmm(P.OP_NONE, alpha, A, d, order, B, n, ldb, beta, C, ldc, kid=kid) == 0
            got = C.reshape(n, m).T if oname == "col" else C.reshape(m, n)
            for psz in (4, 8):
                for fused in (True, False):
                    with oracle.contract(fused):
                        st, Ck = oracle.dcsrmm_kt(oname, psz, alpha, 0, v, ci, rp, m, B, n, ldb, beta, C0, ldc)
                    assert st == 0
                    ref = Ck.reshape(n, m).T if oname == "col" else Ck.reshape(m, n)
                    assert np.all(np.abs(got - ref) <= bound + 1e-300), (oname, kid, psz, fused)
                    # a pinned kid reproduces the KT kernel the reference dispatches for it (csrmm.hpp:779-833), bit for bit
                    # in the fused build: kid 1 / 2 -> 256-bit vectors (4 lanes), kid 3 -> 512-bit (8 lanes)
                    if fused and psz == {1: 4, 2: 4, 3: 8}.get(kid):
                        assert np.array_equal(got, ref), (oname, kid, psz)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n", [32, 36, 48, 100, 128, 130, 256])
def test_csrmm_kid_row_major_on_the_tuned_kernels(n, dtype):
    """Row-major aoclsparse_?csrmm_kid 1 / 2 / 3 runs on the tuned kernels (tile kernel below 128 columns, row-per-wave kernel
    from 128) in their csrmm_row_kt arithmetic: c = c * beta, then c = fma(alpha * a_k, b_kj, c) for the columns the vectors
    cover and c = fma(a_k * b_kj, alpha, c) for the last n mod width ones (csrmm_kt.cpp:244-356) -- bit for bit against the
    restatement pinned on the reference's templates.  Matrices: a 5-point stencil (the benchmark pattern), a random CSR with empty rows, and one with a row longer than
    an LDS tile."""
    import __graft_entry__ as entry
    mats = []
    m, rp, ci, v = entry.laplace5(60)
    mats.append(("laplace", m, m, rp, ci, v))
    rp, ci, v = random_csr(5, 900, 700, lambda r, i: 0 if i % 7 == 3 else r.integers(1, 30))
    mats.append(("random", 900, 700, rp, ci, v))
    rp, ci, v = random_csr(6, 300, 1500, lambda r, i: 1400 if i == 17 else r.integers(0, 12))
    mats.append(("long row", 300, 1500, rp, ci, v))
    # a matrix WITH row groups (3 dofs per mesh node): the group kernels have no KT form, the per-row kernels serve the kid
    from test_gpu_trsv_blocks import node_mesh
    mg, rpg, cig, vg = node_mesh(77, 400, 20, np.full(400, 3))
    mats.append(("row groups", mg, mg, rpg, cig, vg))
    rng = np.random.default_rng(44)
    d = P.Descr()
    for name, m, k, rp, ci, v in mats:
        v = v.astype(dtype)
        A = P.Matrix(0, m, k, rp, ci, v)
        B = rng.uniform(-1, 1, k * n).astype(dtype)
        C0 = rng.uniform(-1, 1, m * n).astype(dtype)
        for alpha, beta in ((1.0, 0.0), (-1.75, 0.625)):
            for kid in (1, 2, 3):
                lanes = kt_lanes(kid, dtype)  # (n % lanes != 0: the last columns take csrmm_row_kt's scalar statement)
                C = C0.copy()
                fn = P.dcsrmm if dtype == np.float64 else P.scsrmm
                assert fn(P.OP_NONE, alpha, A, d, P.ORDER_ROW, B, n, n, beta, C, n, kid=kid) == 0
                kt = oracle.dcsrmm_kt if dtype == np.float64 else oracle.scsrmm_kt
                st, ref = kt("row", lanes, alpha, 0, v, ci, rp, m, B, n, n, beta, C0, n)
                assert st == 0
                u = np.uint64 if dtype == np.float64 else np.uint32
                assert np.array_equal(C.view(u), ref.view(u)), (name, n, kid, alpha, beta, float(np.abs(C - ref).max()))


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_csrmm_kid_column_major_short_rows_use_the_tuned_kernels(dtype):
    """Column-major aoclsparse_?csrmm_kid with every row shorter than the KT vector width (kid 3 on a 5-point stencil; kid 1 / 2
    on rows of <= 3 entries): csrmm_col_kt then never forms a full group and its arithmetic is the kid-0 chain, so the call is
    routed to the tuned column-major kernels -- still bit for bit the KT restatement."""
    import __graft_entry__ as entry
    m, rp, ci, v = entry.laplace5(70)
    rp3, ci3, v3 = random_csr(9, 2000, 1800, lambda r, i: r.integers(0, 4))
    rng = np.random.default_rng(2)
    d = P.Descr()
    u = np.uint64 if dtype == np.float64 else np.uint32
    for name, mm, kk, rp_, ci_, v_, kids in (("stencil", m, m, rp, ci, v, (3,)), ("short", 2000, 1800, rp3, ci3, v3, (1, 2, 3))):
        v_ = v_.astype(dtype)
        A = P.Matrix(0, mm, kk, rp_, ci_, v_)
        for n in (32, 256):
            B = rng.uniform(-1, 1, kk * n).astype(dtype)
            C0 = rng.uniform(-1, 1, mm * n).astype(dtype)
            for alpha, beta in ((1.0, 0.0), (1.5, -0.25)):
                for kid in kids:
                    lanes = kt_lanes(kid, dtype)
                    if dtype == np.float32 and name == "stencil" and lanes <= 5:
                        continue
                    C = C0.copy()
                    fn = P.dcsrmm if dtype == np.float64 else P.scsrmm
                    assert fn(P.OP_NONE, alpha, A, d, P.ORDER_COLUMN, B, n, kk, beta, C, mm, kid=kid) == 0
                    kt = oracle.dcsrmm_kt if dtype == np.float64 else oracle.scsrmm_kt
                    st, ref = kt("col", lanes, alpha, 0, v_, ci_, rp_, mm, B, n, kk, beta, C0, mm)
                    assert st == 0 and np.array_equal(C.view(u), ref.view(u)), (name, n, kid, alpha)


@pytest.mark.parametrize("n", [7, 16, 35])
def test_float_csrmm_kid_reproduces_the_kt_orders(n):
    """aoclsparse_scsrmm_kid: kid 1 / 2 -> the 8-lane (256-bit) float KT kernels, kid 3 -> the 16-lane (512-bit) ones, both layouts,
    bit for bit against the restatement pinned on the reference's own templates (tests/golden/kt_vectors.json: csrmm_col_s,
    csrmm_row_s); every kid within (len + 4) eps32 |alpha| sum |a b| + 3 eps32 |beta c| of every order."""
    f32 = np.float32
    eps32 = float(np.finfo(f32).eps)
    m, k = 400, 350
    rp, ci, v = random_csr(91, m, k, lambda r, i: r.integers(0, 50))
    v = v.astype(f32)
    A = P.Matrix(0, m, k, rp, ci, v)  # float32 values -> aoclsparse_create_scsr
    d = P.Descr()
    rng = np.random.default_rng(19)
    lens = np.diff(rp)
    absA = dense_of(m, k, 0, rp, ci, np.abs(v).astype(np.float64))
    for alpha, beta in ((f32(1.0), f32(0.0)), (f32(-2.5), f32(0.75))):
        for order, oname in ((P.ORDER_COLUMN, "col"), (P.ORDER_ROW, "row")):
            ldb, ldc = (k, m) if oname == "col" else (n, n)
            B = rng.uniform(-1, 1, k * n).astype(f32)
            C0 = rng.uniform(-1, 1, m * n).astype(f32)
            Bm = B.reshape(n, k).T if oname == "col" else B.reshape(k, n)
            Cm = C0.reshape(n, m).T if oname == "col" else C0.reshape(m, n)
            bound = ((lens[:, None] + 4) * eps32 * abs(float(alpha)) * (absA @ np.abs(Bm.astype(np.float64)))
                     + 3 * eps32 * np.abs(float(beta) * Cm.astype(np.float64)))
            for kid in (0, 1, 2, 3):
                C = C0.copy()
                assert P.scsrmm(P.OP_NONE, float(alpha), A, d, order, B, n, ldb, float(beta), C, ldc, kid=kid) == 0
                got = C.reshape(n, m).T if oname == "col" else C.reshape(m, n)
                for psz in (8, 16):
                    st, Ck = oracle.scsrmm_kt(oname, psz, float(alpha), 0, v, ci, rp, m, B, n, ldb, float(beta), C0, ldc)
                    assert st == 0
                    ref = Ck.reshape(n, m).T if oname == "col" else Ck.reshape(m, n)
                    assert np.all(np.abs(got.astype(np.float64) - ref.astype(np.float64)) <= bound + 1e-30), (oname, kid, psz)
                    if psz == {1: 8, 2: 8, 3: 16}.get(kid):
                        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), (oname, kid, psz, float(alpha))


def test_spmv_within_bound_of_the_gcc_build_orders():
    """The GPU SpMV reproduces the FUSED build of the reference bit for bit (test_gpu_parity.py).  Against the GCC -march=znver2
    build, whose scalar loops round the product and the sum separately (oracle.c header), the stated tolerance is
    (len + 2) eps sum |a_ij x_j| per row."""
    m, n = 3000, 2500
    rp, ci, v = random_csr(5, m, n, lambda r, i: r.integers(0, 60))
    x = np.random.default_rng(6).uniform(-1, 1, n)
    A = P.Matrix(0, m, n, rp, ci, v)
    d = P.Descr()
    lens = np.diff(rp)
    scale = dense_of(m, n, 0, rp, ci, np.abs(v)) @ np.abs(x)
    for kid in (0, 1, 3):
        assert L.aoclsparse_set_mv_hint_kid(A.h, P.OP_NONE, d.h, 0, kid) == 0
        y = np.zeros(m)
        assert P.dmv(P.OP_NONE, 1.0, A, d, x, 0.0, y) == 0
        with oracle.contract(True):
            st, yf = oracle.dcsrmv_order({0: "ref", 1: "lane4", 3: "lane8"}[kid], 0, 1.0, m, v, ci, rp, x, 0.0, np.zeros(m))
        with oracle.contract(False):
            st, yg = oracle.dcsrmv_order({0: "ref", 1: "lane4", 3: "lane8"}[kid], 0, 1.0, m, v, ci, rp, x, 0.0, np.zeros(m))
        nnz = int(rp[-1])
        if nnz > 10 * m or kid == 0:   # the reference overrides the kid to 0 when nnz <= 10 m (csrmv.hpp:322-355)
            assert np.array_equal(y, yf), kid
        assert np.all(np.abs(y - yg) <= (lens + 2) * EPS64 * scale + 1e-300), kid


# ---------------------------------------------------------------------------------------------------------------------------
# in-library multi-device csrmm (one process, one worker thread + runtime slot + handle replica per device)
# ---------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("order", ["col", "row"])
def test_csrmm_multi_device_entry_points_on_one_gpu(order):
    """aoclsparse_mi355_dcsrmm_multi / _multi_slabs with every slot on device 0 (what a one-GPU box can run: the slots still
    have their own streams, staging buffers and handle replicas, so the whole N-device control flow executes): the result
    equals the single-call product bit for bit for 1, 2, 3 and 8 slots, both layouts, beta classes, a column count that
    leaves a slot without columns; value updates reach the replicas; argument errors."""
    m, k, n = 3000, 2600, 40
    rp, ci, v = random_csr(303, m, k, lambda r, i: r.integers(0, 12))
    A = P.Matrix(0, m, k, rp, ci, v)
    d = P.Descr()
    assert L.aoclsparse_set_mm_hint(A.h, P.OP_NONE, d.h, 10) == 0 and L.aoclsparse_optimize(A.h) == 0
    rng = np.random.default_rng(8)
    colmaj = order == "col"
    o = P.ORDER_COLUMN if colmaj else P.ORDER_ROW
    ldb, ldc = (k, m) if colmaj else (n, n)
    B = rng.uniform(-1, 1, k * n)
    C0 = rng.uniform(-1, 1, m * n)
    st, dev0, _, _ = P.device_info()
    assert st == 0
    for alpha, beta in ((1.0, 0.0), (-0.75, 1.5)):
        ref = C0.copy()
        assert P.dcsrmm(P.OP_NONE, alpha, A, d, o, B, n, ldb, beta, ref, ldc) == 0
        for ndev in (1, 2, 3, 8, 12):   # 40 columns / 12 slots: some slots own no 4-column block
            C = C0.copy()
            assert P.dcsrmm_multi(P.OP_NONE, alpha, A, d, o, B, n, ldb, beta, C, ldc, [dev0] * ndev) == 0, ndev
            assert np.array_equal(C, ref), (ndev, alpha, beta)
        # the handle was optimized before its first multi-device call: every replica is a device-to-device copy of its device
        # format ("A replicated in its device format once"), none re-analysed the host arrays
        assert L.aoclsparse_mi355_replica_count(A.h) == 11 and L.aoclsparse_mi355_replicas_cloned(A.h) == 11
        # slabs resident on the device(s)
        ndev = 3
        shards = [P.column_shard(n, ndev, r) for r in range(ndev)]
        Bm = B.reshape(n, k) if colmaj else B.reshape(k, n)
        Cm = C0.reshape(n, m) if colmaj else C0.reshape(m, n)
        Bs = [dev(np.ascontiguousarray(Bm[j0:j1] if colmaj else Bm[:, j0:j1])) for j0, j1 in shards]
        Cs = [dev(np.ascontiguousarray(Cm[j0:j1] if colmaj else Cm[:, j0:j1])) for j0, j1 in shards]
        lb = [ldb if colmaj else max(j1 - j0, 1) for j0, j1 in shards]
        assert len(set(lb)) == 1 or not colmaj
        if colmaj:
            assert P.dcsrmm_multi_slabs(P.OP_NONE, alpha, A, d, o, Bs, n, ldb, beta, Cs, ldc, [dev0] * ndev) == 0
            R = ref.reshape(n, m)
            for (j0, j1), c in zip(shards, Cs):
                assert np.array_equal(c.cpu().numpy(), R[j0:j1])
    # the replicas follow value updates (they are dropped and rebuilt)
    v2 = v * 1.5
    assert L.aoclsparse_dupdate_values(A.h, len(v2), P._ptr(np.ascontiguousarray(v2))) == 0
    ref = C0.copy()
    assert P.dcsrmm(P.OP_NONE, 1.0, A, d, o, B, n, ldb, 0.0, ref, ldc) == 0
    C = C0.copy()
    assert P.dcsrmm_multi(P.OP_NONE, 1.0, A, d, o, B, n, ldb, 0.0, C, ldc, [dev0, dev0]) == 0
    assert np.array_equal(C, ref)
    so, Cr = oracle.dcsrmm("col", 1.0, 0, v2, ci, rp, m, B if colmaj else np.ascontiguousarray(B.reshape(k, n).T).ravel(), n, k,
                           0.0, C0 if colmaj else np.ascontiguousarray(C0.reshape(m, n).T).ravel(), m)
    got = C if colmaj else np.ascontiguousarray(C.reshape(m, n).T).ravel()
    assert np.array_equal(got, Cr)
    assert L.aoclsparse_mi355_replica_count(A.h) == 1  # dropped by the value update, one rebuilt by the two-slot call
    # a handle that was never optimized nor used: its replicas are built whichever way the race with slot 0 goes (slot 0
    # uploads the primary copy while the workers start) -- the result is the same
    A2 = P.Matrix(0, m, k, rp, ci, v2)
    C = C0.copy()
    assert P.dcsrmm_multi(P.OP_NONE, 1.0, A2, d, o, B, n, ldb, 0.0, C, ldc, [dev0] * 4) == 0
    assert np.array_equal(C, ref) and 0 <= L.aoclsparse_mi355_replicas_cloned(A2.h) <= 3
    # argument errors
    assert P.dcsrmm_multi(P.OP_NONE, 1.0, A, d, o, B, n, ldb, 0.0, C, ldc, [dev0 + 1]) != 0      # slot 0 must be the library's device
    assert L.aoclsparse_mi355_dcsrmm_multi(P.OP_NONE, 1.0, A.h, d.h, o, P._ptr(B), n, ldb, 0.0, P._ptr(C), ldc, 0, None) != 0
    assert P.dcsrmm_multi(P.OP_NONE, 1.0, A, d, o, B, n, ldb, 0.0, C, ldc, [dev0, 4096]) != 0   # no such device


def test_csrmm_env_devices_routes_host_operands():
    """AOCLSPARSE_MI355_DEVICES=N: plain aoclsparse_dcsrmm with host operands takes the multi-device path (here N slots on the
    only device) and returns the same bits; device operands keep the single-device path."""
    import subprocess
    import sys
    import textwrap
    code = textwrap.dedent("""
        import sys, numpy as np
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        from util import pkg, random_csr
        import oracle
        P = pkg()
        m, k, n = 2000, 1800, 64
        rp, ci, v = random_csr(5, m, k, lambda r, i: r.integers(0, 9))
        A = P.Matrix(0, m, k, rp, ci, v); d = P.Descr()
        rng = np.random.default_rng(1); B = rng.uniform(-1, 1, k * n); C = np.zeros(m * n)
        assert P.dcsrmm(P.OP_NONE, 2.0, A, d, P.ORDER_COLUMN, B, n, k, 0.0, C, m) == 0
        so, Cr = oracle.dcsrmm("col", 2.0, 0, v, ci, rp, m, B, n, k, 0.0, np.zeros(m * n), m)
        assert np.array_equal(C, Cr)
        print("replicas", P.lib().aoclsparse_mi355_replica_count(A.h))
    """) % (HERE, os.path.dirname(HERE))
    for nd, want in (("1", "replicas 0"), ("3", "replicas 2")):
        env = dict(os.environ, AOCLSPARSE_MI355_DEVICES=nd)
        out1 = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert out1.returncode == 0 and want in out1.stdout, (nd, out1.stdout, out1.stderr[-2000:])
