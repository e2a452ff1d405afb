"""Round-5 GPU tests: the one-launch merge-path kernel under residency pressure and uneven load (tools/stress_mergepath.py in small),
and the differential fuzzer of the round's kernels (tools/fuzz_r5.py) as a short run.  The other round-5 changes are tested where
their subjects already were: test_gpu_parity.py (tree / strict modes, merge-path), test_gpu_configs.py (full-size mix in both modes),
test_gpu_r4.py (blocked-ELL with Inf / NaN, sp2m finalize)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(tool, *args, timeout=600):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool)] + [str(a) for a in args], capture_output=True, text=True,
                       timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])


def test_merge_path_look_back_under_residency_pressure_and_uneven_load():
    """~9,000 tiles of 1,024 items against the 2,048 workgroups the chip holds, three rows that cross ~490 tiles each, ten launches
    (every second one next to a competing stream that keeps all CUs busy): no look-back expires (no NaN), every row within the
    bound, and every launch returns the same bits (the order of additions depends on the tiling only)."""
    out = _run("stress_mergepath.py", 2000000, 3, 500000, 10)
    assert out["tiles"] > 4 * 2048 and out["tiles_per_long_row"] >= 400
    assert out["every_launch_bit_identical"] and out["worst_err_over_bound"] <= 1.0 and out["with_competing_stream"] == 5


def test_round5_differential_fuzz_short():
    """tree / merge-path / strict SpMV, blocked-ELL csrmm with Inf / NaN, device csr2csc with long rows: tools/fuzz_r5.py, 25 cases"""
    out = _run("fuzz_r5.py", 25, 3)
    assert not any(out["mismatches"].values()), out
    assert out["checked"]["spmv"] >= 30 and out["checked"]["bell"] >= 10 and out["checked"]["csr2csc"] >= 2


def test_release_staging_reaches_the_secondary_runtime_slots():
    """aoclsparse_mi355_release_staging is process-wide: the slabs the slots of a multi-device call staged for their
    share of host operands are freed too (here: four slots on device 0), counted in bytes_freed, and the next multi-device call
    allocates them again and returns the same bits."""
    import ctypes
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import pkg, random_csr
    P = pkg()
    L = P.lib()
    m, k, n = 4000, 3000, 64
    rp, ci, v = random_csr(77, m, k, lambda r, i: r.integers(0, 9))
    A = P.Matrix(0, m, k, rp, ci, v)
    d = P.Descr()
    assert L.aoclsparse_set_mm_hint(A.h, P.OP_NONE, d.h, 10) == 0 and L.aoclsparse_optimize(A.h) == 0
    rng = np.random.default_rng(3)
    B, C0 = rng.uniform(-1, 1, k * n), rng.uniform(-1, 1, m * n)
    st, dev0, _, _ = P.device_info()
    assert st == 0
    freed = ctypes.c_size_t(0)
    assert L.aoclsparse_mi355_release_staging(ctypes.byref(freed)) == 0  # start from nothing staged
    C1 = C0.copy()
    assert P.dcsrmm_multi(P.OP_NONE, 1.5, A, d, P.ORDER_ROW, B, n, n, -0.5, C1, n, [dev0] * 4) == 0
    torch.cuda.synchronize()
    assert L.aoclsparse_mi355_release_staging(ctypes.byref(freed)) == 0
    # each of the four slots staged its 16-column slab of B (k x 16) and C (m x 16): at least those bytes come back
    assert freed.value >= 8 * 16 * (k + m) * 3, freed.value
    assert L.aoclsparse_mi355_release_staging(ctypes.byref(freed)) == 0 and freed.value == 0
    C2 = C0.copy()
    assert P.dcsrmm_multi(P.OP_NONE, 1.5, A, d, P.ORDER_ROW, B, n, n, -0.5, C2, n, [dev0] * 4) == 0
    assert np.array_equal(C1, C2)


def test_exported_cxx_instantiations_compute_on_the_gpu(tmp_path):
    """A C++ program that only DECLARES aoclsparse::create_csr / mv / trsv / sp2m (what the reference's public header gives a caller)
    links against the library's exported instantiations through the versioned name and gets the right numbers from the GPU:
    y = 2 A x - y on the 5-point Laplacian, the unit-lower solve of its strict lower triangle, and nnz(A A)."""
    src = tmp_path / "cxx_abi.cpp"
    src.write_text(r"""
#include "aoclsparse.h"
#include <cmath>
#include <cstdio>
#include <vector>
namespace aoclsparse {
template <typename T> aoclsparse_status mv(aoclsparse_operation, const T *, aoclsparse_matrix, const aoclsparse_mat_descr,
                                           const T *, const T *, T *);
template <typename T> aoclsparse_status create_csr(aoclsparse_matrix *, aoclsparse_index_base, aoclsparse_int, aoclsparse_int,
                                                   aoclsparse_int, aoclsparse_int *, aoclsparse_int *, T *, bool = false);
template <typename T> aoclsparse_status trsv(const aoclsparse_operation, const T, aoclsparse_matrix, const aoclsparse_mat_descr,
                                             const T *, const aoclsparse_int, T *, const aoclsparse_int, aoclsparse_int = -1);
template <typename T> aoclsparse_status sp2m(aoclsparse_operation, const aoclsparse_mat_descr, const aoclsparse_matrix,
                                             aoclsparse_operation, const aoclsparse_mat_descr, const aoclsparse_matrix,
                                             aoclsparse_request, aoclsparse_matrix *);
}
int main() {
    const int g = 60, m = g * g;
    std::vector<aoclsparse_int> rp(m + 1, 0), ci;
    std::vector<double> v;
    for(int r = 0; r < m; r++) {
        const int i = r / g, j = r % g;
        const int cand[5] = {r - g, r - 1, r, r + 1, r + g};
        const bool ok[5] = {i > 0, j > 0, true, j < g - 1, i < g - 1};
        for(int k = 0; k < 5; k++) if(ok[k]) { ci.push_back(cand[k]); v.push_back(k == 2 ? 4.0 : -1.0); }
        rp[r + 1] = (aoclsparse_int)ci.size();
    }
    aoclsparse_matrix A = nullptr;
    if(aoclsparse::create_csr<double>(&A, aoclsparse_index_base_zero, m, m, (aoclsparse_int)v.size(), rp.data(), ci.data(), v.data()) != aoclsparse_status_success) return 1;
    aoclsparse_mat_descr d = nullptr;
    if(aoclsparse_create_mat_descr(&d) != aoclsparse_status_success) return 2;
    std::vector<double> x(m), y(m, 1.0), yr(m);
    for(int r = 0; r < m; r++) x[r] = std::sin(0.01 * r);
    for(int r = 0; r < m; r++) { double s = 0.0; for(int p = rp[r]; p < rp[r + 1]; p++) s = std::fma(v[p], x[ci[p]], s); yr[r] = std::fma(-1.0, 1.0, 2.0 * s); }
    const double alpha = 2.0, beta = -1.0;
    if(aoclsparse::mv<double>(aoclsparse_operation_none, &alpha, A, d, x.data(), &beta, y.data()) != aoclsparse_status_success) return 3;
    for(int r = 0; r < m; r++) if(y[r] != yr[r]) { std::printf("mv row %d: %.17g vs %.17g\n", r, y[r], yr[r]); return 4; }
    // unit-lower solve: x = L^{-1} b with L = I + strict lower triangle of A, b = L * 1
    aoclsparse_set_mat_type(d, aoclsparse_matrix_type_triangular);
    aoclsparse_set_mat_fill_mode(d, aoclsparse_fill_mode_lower);
    aoclsparse_set_mat_diag_type(d, aoclsparse_diag_type_unit);
    std::vector<double> b(m), s(m);
    for(int r = 0; r < m; r++) { double t = 1.0; for(int p = rp[r]; p < rp[r + 1]; p++) if(ci[p] < r) t += v[p]; b[r] = t; }
    if(aoclsparse::trsv<double>(aoclsparse_operation_none, 1.0, A, d, b.data(), 1, s.data(), 1) != aoclsparse_status_success) return 5;
    for(int r = 0; r < m; r++) if(std::fabs(s[r] - 1.0) > 1e-9) { std::printf("trsv row %d: %.17g\n", r, s[r]); return 6; }
    // C = A * A
    aoclsparse_set_mat_type(d, aoclsparse_matrix_type_general);
    aoclsparse_matrix C = nullptr;
    if(aoclsparse::sp2m<double>(aoclsparse_operation_none, d, A, aoclsparse_operation_none, d, A, aoclsparse_stage_full_computation, &C) != aoclsparse_status_success) return 7;
    aoclsparse_index_base cb; aoclsparse_int cm, cn, cnnz, *crp, *cci; double *cv;
    if(aoclsparse_export_dcsr(C, &cb, &cm, &cn, &cnnz, &crp, &cci, &cv) != aoclsparse_status_success) return 8;
    // row r of A*A touches the 13-point diamond clipped by the grid: count it directly
    long want = 0;
    for(int i = 0; i < g; i++) for(int j = 0; j < g; j++)
        for(int di = -2; di <= 2; di++) for(int dj = -2; dj <= 2; dj++)
            if(std::abs(di) + std::abs(dj) <= 2 && i + di >= 0 && i + di < g && j + dj >= 0 && j + dj < g) want++;
    if(cm != m || cn != m || cnnz != want) { std::printf("sp2m nnz %d vs %ld\n", (int)cnnz, want); return 9; }
    std::printf("ok %d\n", (int)cnnz);
    return (aoclsparse_destroy(&C) == aoclsparse_status_success && aoclsparse_destroy(&A) == aoclsparse_status_success
            && aoclsparse_destroy_mat_descr(d) == aoclsparse_status_success) ? 0 : 10;
}
""")
    exe = tmp_path / "cxx_abi"
    libdir = os.path.join(ROOT, "aocl-sparse_amd", "lib")
    subprocess.check_call(["g++", "-std=c++17", "-O1", str(src), "-I" + os.path.join(ROOT, "include"), "-L" + libdir, "-laoclsparse",
                           "-Wl,-rpath," + libdir, "-o", str(exe)])
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith("ok"), (r.returncode, r.stdout[-500:], r.stderr[-500:])


def test_trsv_solves_of_one_handle_on_alternating_streams():
    """The sync-free solve's workspaces belong to the handle and are reused in stream order; a solve on another stream than the
    previous one must not start before that one is done (the library waits for the device then -- never for the old stream, which
    may no longer exist).  Solves of one handle alternate between two streams with different right-hand sides and no
    synchronisation by the caller; one of the streams is then destroyed through the HIP runtime the library is bound to and
    the handle solves on."""
    import ctypes
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    from util import pkg, triangular_system
    P = pkg()
    L = P.lib()
    st, path = P.hip_runtime_path()
    assert st == 0
    hip = ctypes.CDLL(path)
    hip.hipStreamCreate.argtypes, hip.hipStreamDestroy.argtypes = [ctypes.POINTER(ctypes.c_void_p)], [ctypes.c_void_p]
    hip.hipStreamSynchronize.argtypes = [ctypes.c_void_p]
    m = 60000
    rp, ci, v = triangular_system(91, m, 3, band=300)
    A = P.Matrix(0, m, m, rp, ci, v)
    dl = P.Descr(mtype=P.TYPE_TRIANGULAR, fill=P.FILL_LOWER, diag=P.DIAG_NON_UNIT)
    assert L.aoclsparse_set_sv_hint(A.h, P.OP_NONE, dl.h, 100) == 0 and L.aoclsparse_optimize(A.h) == 0
    o = oracle.dcsr_optimize(m, m, len(v), 0, rp, ci, v)
    bs = [np.random.default_rng(50 + k).uniform(-1, 1, m) for k in range(2)]
    ref = [oracle.dtrsv("l", 1.0, m, 0, o["val"], o["ind"], o["ptr"], o["idiag"], b, False)[1] for b in bs]
    bd = [torch.from_numpy(b).cuda() for b in bs]
    xd = [torch.zeros(m, dtype=torch.float64, device="cuda") for _ in range(2)]
    torch.cuda.synchronize()
    L.aoclsparse_mi355_set_pointer_mode(P.PTR_DEVICE)
    hs = []
    try:
        for _ in range(2):
            h = ctypes.c_void_p()
            assert hip.hipStreamCreate(ctypes.byref(h)) == 0
            hs.append(h)
        for it in range(6):
            for k in range(2):
                assert L.aoclsparse_mi355_set_stream(hs[k]) == 0
                assert P.dtrsv(P.OP_NONE, 1.0, A, dl, bd[k], xd[k]) == 0
        for k in range(2):
            assert hip.hipStreamSynchronize(hs[k]) == 0
            assert np.array_equal(xd[k].cpu().numpy(), ref[k])
        assert L.aoclsparse_mi355_set_stream(hs[0]) == 0
        assert hip.hipStreamDestroy(hs[1]) == 0
        hs[1] = None
        xd[0].zero_()
        torch.cuda.synchronize()
        assert P.dtrsv(P.OP_NONE, 1.0, A, dl, bd[0], xd[0]) == 0
        assert hip.hipStreamSynchronize(hs[0]) == 0 and np.array_equal(xd[0].cpu().numpy(), ref[0])
    finally:
        assert L.aoclsparse_mi355_set_stream(None) == 0
        L.aoclsparse_mi355_set_pointer_mode(P.PTR_AUTO)
        torch.cuda.synchronize()
        for h in hs:
            if h is not None:
                hip.hipStreamDestroy(h)


def test_extreme_value_known_answers():
    """The reference's extreme-value vectors (tests/unit_tests/extreme_value_tests.cpp, fixture reference_kats_r5.json) through
    the library: aoclsparse_sp2m (one and two stages), aoclsparse_dadd, aoclsparse_dcsrmm (no kid, kid 1, kid 3; host and
    device operands) and aoclsparse_ddoti -- NaN where the reference has NaN, the same infinities, finite values within its
    one-ulp-scale tolerance; and bit for bit what the oracle returns for the same inputs."""
    import ctypes
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    from util import pkg
    from test_oracle_golden_r5 import ev_array, ev_match, sorted_csr
    from test_gpu_parity_r3 import _export
    P = pkg()
    L = P.lib()
    with open(os.path.join(ROOT, "tests", "golden", "reference_kats_r5.json")) as f:
        k5 = json.load(f)
    I = k5["init"]
    m = I["m"]
    a = (np.array(I["A_row_ptr"], np.int32), np.array(I["A_col_ind"], np.int32), ev_array(I["A_val"]))
    b = (np.array(I["B_row_ptr"], np.int32), np.array(I["B_col_ind"], np.int32), ev_array(I["B_val"]))
    A, B, d = P.Matrix(0, m, m, *a), P.Matrix(0, m, m, *b), P.Descr()
    same = lambda x, y: np.array_equal(np.asarray(x).view(np.uint64) if np.asarray(x).dtype == np.float64 else x,
                                       np.asarray(y).view(np.uint64) if np.asarray(y).dtype == np.float64 else y)
    # sp2m, one stage and two
    so, pc, ic, vc = oracle.dcsr2m(m, m, 0, a[0], a[1], a[2], 0, b[0], b[1], b[2])
    for stages in ((P.STAGE_FULL,), (P.STAGE_NNZ_COUNT, P.STAGE_FINALIZE)):
        C = ctypes.c_void_p()
        for stg in stages:
            assert L.aoclsparse_sp2m(P.OP_NONE, d.h, A.h, P.OP_NONE, d.h, B.h, stg, ctypes.byref(C)) == 0
        _, cm, cn, cz, row, col, val = _export(C)
        assert (cm, cn, cz) == (m, m, 34) and np.array_equal(row, pc) and np.array_equal(col, ic)
        assert np.array_equal(np.isnan(val), np.isnan(vc)) and same(val[~np.isnan(vc)], vc[~np.isnan(vc)])
        ev_match(sorted_csr(row, col, val)[1], ev_array(k5["sp2m"]["C_exp_val"]))
        assert L.aoclsparse_destroy(ctypes.byref(C)) == 0
    # add
    C = ctypes.c_void_p()
    assert L.aoclsparse_dadd(P.OP_NONE, A.h, 1.0, B.h, ctypes.byref(C)) == 0
    _, cm, cn, cz, row, col, val = _export(C)
    assert list(row) == k5["add"]["C_exp_row_ptr"] and list(sorted_csr(row, col, val)[0]) == k5["add"]["C_exp_col_ind"]
    ev_match(sorted_csr(row, col, val)[1], ev_array(k5["add"]["C_exp_val"]))
    assert L.aoclsparse_destroy(ctypes.byref(C)) == 0
    # csrmm: row-major, alpha 1, beta 0, C zero on entry (the reference resizes it) -- every kid, host and device operands
    Bd, exp = ev_array(I["B_dense_row_major"]), ev_array(k5["csrmm"]["C_exp_val"])
    for kid, lanes in ((None, 0), (1, 4), (3, 8), (0, 0)):
        st, Co = (oracle.dcsrmm_kt("row", lanes, 1.0, 0, a[2], a[1], a[0], m, Bd, m, m, 0.0, np.zeros(m * m), m) if lanes
                  else oracle.dcsrmm("row", 1.0, 0, a[2], a[1], a[0], m, Bd, m, m, 0.0, np.zeros(m * m), m))
        for on_device in (False, True):
            Ch = np.zeros(m * m)
            Bx, Cx = (torch.from_numpy(Bd).cuda(), torch.from_numpy(Ch).cuda()) if on_device else (Bd, Ch)
            assert P.dcsrmm(P.OP_NONE, 1.0, A, d, P.ORDER_ROW, Bx, m, m, 0.0, Cx, m, kid=kid) == 0
            torch.cuda.synchronize()
            got = Cx.cpu().numpy() if on_device else Ch
            ev_match(got, exp)
            assert np.array_equal(np.isnan(got), np.isnan(Co)) and same(got[~np.isnan(Co)], Co[~np.isnan(Co)]), (kid, on_device)
    # sparse dot
    D = k5["dot"]
    indx, y = np.array(D["indx"], np.int32), np.array(D["y"], np.float64)
    for case in D["cases"]:
        x = np.array(D["x"], np.float64)
        x[0], x[1] = ev_array([case["x0"]])[0], ev_array([case["x1"]])[0]
        got = L.aoclsparse_ddoti(len(indx), P._ptr(x), P._ptr(indx), P._ptr(y))
        ev_match(np.array([got]), ev_array([case["expected"]]))


def test_symmetric_mv_matches_the_matrix_the_reference_optimize_builds():
    """optimize_symm_herm_tests.cpp:39-938 through the library: for the reference's four matrices (fixture symm_opt), both
    triangles, the three diagonal types, both bases and op = none / transpose, aoclsparse_dmv after set_mv_hint + optimize returns
    E x -- E being the symmetric matrix the reference lists as the content of its optimized copy (small integers: exact)."""
    import numpy as np
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import pkg
    from test_oracle_golden_r5 import symm_expected_y
    P = pkg()
    L = P.lib()
    with open(os.path.join(ROOT, "tests", "golden", "reference_kats_r5.json")) as f:
        k5 = json.load(f)
    ran = 0
    for M in k5["symm_opt"]["matrices"]:
        m = M["m"]
        x = np.arange(1, m + 1, dtype=np.float64) * np.array([1, -2, 3, 5][:m])
        for base in (0, 1):
            rp, ci, v = np.array(M["row_ptr"], np.int32) + base, np.array(M["col_ind"], np.int32) + base, np.array(M["val"])
            for fill, tri in ((P.FILL_LOWER, "lower"), (P.FILL_UPPER, "upper")):
                for diag in (P.DIAG_NON_UNIT, P.DIAG_UNIT, P.DIAG_ZERO):
                    for op in (P.OP_NONE, P.OP_TRANSPOSE):
                        A = P.Matrix(base, m, m, rp, ci, v)
                        d = P.Descr(mtype=P.TYPE_SYMMETRIC, fill=fill, diag=diag, base=base)
                        assert L.aoclsparse_set_mv_hint(A.h, op, d.h, 1) == 0 and L.aoclsparse_optimize(A.h) == 0
                        y = np.full(m, 7.0)
                        assert P.dmv(op, 1.0, A, d, x, 0.0, y) == 0
                        assert np.array_equal(y, symm_expected_y(M["expected"][tri], diag, x)), (M["id"], base, tri, diag, op)
                        ran += 1
    assert ran == 4 * 2 * 2 * 3 * 2


def test_consecutive_products_alternate_the_sweep_direction_with_the_same_bits():
    """Odd-numbered products of a handle walk the SELL-64 slices (and the row blocks of the CSR kernel) in DESCENDING order, so
    that what one product leaves in the Infinity Cache is where the next one starts.  The direction is not allowed to show in
    the result: six consecutive products return the oracle's bits, on the short-row SELL kernel (5,625 slices), the general
    PACK-4 kernel (8-lane order), the CSR-Adaptive kernel of an un-hinted handle (before its promotion), float and complex."""
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    from util import banded_rows, laplace5, pkg
    P = pkg()
    L = P.lib()
    d = P.Descr()
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()

    def six(run, ref, what):
        for it in range(6):
            got = run()
            assert np.array_equal(got, ref), (what, it, float(np.max(np.abs(got - ref))))

    # (a) short-row SELL kernel, shared column lists
    m, rp, ci, v = laplace5(600)
    x = np.sin(0.01 * np.arange(m))
    A = P.Matrix(0, m, m, rp, ci, v)
    assert L.aoclsparse_set_mv_hint(A.h, P.OP_NONE, d.h, 100) == 0 and L.aoclsparse_optimize(A.h) == 0
    assert A.spmv_info().kernel in (3, 4)
    so, yr = oracle.dcsrmv(-1, 0, 1.5, m, len(v), v, ci, rp, x, 0.0, np.zeros(m))
    xd, yd = dev(x), torch.zeros(m, dtype=torch.float64, device="cuda")

    def run_a():
        assert P.dmv(P.OP_NONE, 1.5, A, d, xd, 0.0, yd) == 0
        torch.cuda.synchronize()
        return yd.cpu().numpy()
    six(run_a, yr, "short")
    # (a') the same matrix in float
    vf = v.astype(np.float32)
    Af = P.Matrix(0, m, m, rp, ci, vf)
    assert L.aoclsparse_set_mv_hint(Af.h, P.OP_NONE, d.h, 100) == 0 and L.aoclsparse_optimize(Af.h) == 0
    xf = x.astype(np.float32)
    so, yrf = oracle.scsrmv("lane8", 0, 1.0, m, vf, ci, rp, xf, 0.0, np.zeros(m, np.float32))
    xfd, yfd = dev(xf), torch.zeros(m, dtype=torch.float32, device="cuda")

    def run_f():
        assert P.smv(P.OP_NONE, 1.0, Af, d, xfd, 0.0, yfd) == 0
        torch.cuda.synchronize()
        return yfd.cpu().numpy()
    six(run_f, yrf, "float")
    # (b) general kernel, PACK 4, 8-lane order (nnz > 10 m)
    m2 = 40000
    rp2, ci2, v2 = banded_rows(17, m2, m2, lambda r, i: r.integers(18, 26))
    x2 = np.random.default_rng(6).uniform(-1, 1, m2)
    B = P.Matrix(0, m2, m2, rp2, ci2, v2)
    assert L.aoclsparse_set_mv_hint(B.h, P.OP_NONE, d.h, 100) == 0 and L.aoclsparse_optimize(B.h) == 0
    assert B.spmv_info().kernel in (3, 4) and len(v2) > 10 * m2
    so, yr2 = oracle.dcsrmv(-1, 0, 1.0, m2, len(v2), v2, ci2, rp2, x2, 0.0, np.zeros(m2))
    x2d, y2d = dev(x2), torch.zeros(m2, dtype=torch.float64, device="cuda")

    def run_b():
        assert P.dmv(P.OP_NONE, 1.0, B, d, x2d, 0.0, y2d) == 0
        torch.cuda.synchronize()
        return y2d.cpu().numpy()
    six(run_b, yr2, "pack4")
    # (c) no hint: the row-block kernel (the handle is promoted to SELL-64 at its 8th product only)
    C = P.Matrix(0, m, m, rp, ci, v)

    def run_c():
        assert P.dmv(P.OP_NONE, 1.5, C, d, xd, 0.0, yd) == 0 and C.spmv_info().kernel == 1
        torch.cuda.synchronize()
        return yd.cpu().numpy()
    six(run_c, yr, "csr-adaptive")


def test_consecutive_csrmm_products_alternate_the_block_order_with_the_same_bits():
    """Every second row-major csrmm product of a handle runs its row blocks in descending order (mm_order.hpp).  Five consecutive
    products -- 32 columns (the slab kernel) and 256 columns (row per wavefront; row runs in overwrite mode), C read and
    overwritten -- return the same bits, and the first four columns are the oracle's."""
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    from util import beta0_overwrite, laplace5, pkg
    P = pkg()
    L = P.lib()
    d = P.Descr()
    m, rp, ci, v = laplace5(300)
    A = P.Matrix(0, m, m, rp, ci, v)
    assert L.aoclsparse_set_mm_hint(A.h, P.OP_NONE, d.h, 100) == 0 and L.aoclsparse_optimize(A.h) == 0
    rng = np.random.default_rng(9)
    for n in (32, 256):
        B = rng.uniform(-1, 1, m * n)
        Bd = torch.from_numpy(B).cuda()
        Bc = np.ascontiguousarray(B.reshape(m, n)[:, :4].T).ravel()
        so, Cr = oracle.dcsrmm("col", 1.25, 0, v, ci, rp, m, Bc, 4, m, 0.0, np.zeros(4 * m), m)
        for overwrite in (False, True):
            first = None
            for it in range(5):
                Cd = torch.zeros(m * n, dtype=torch.float64, device="cuda")
                if overwrite:
                    with beta0_overwrite(P):
                        assert P.dcsrmm(P.OP_NONE, 1.25, A, d, P.ORDER_ROW, Bd, n, n, 0.0, Cd, n) == 0
                else:
                    assert P.dcsrmm(P.OP_NONE, 1.25, A, d, P.ORDER_ROW, Bd, n, n, 0.0, Cd, n) == 0
                torch.cuda.synchronize()
                got = Cd.cpu().numpy()
                if first is None:
                    first = got
                    assert np.array_equal(np.ascontiguousarray(got.reshape(m, n)[:, :4].T).ravel(), Cr), (n, overwrite)
                else:
                    assert np.array_equal(got, first), (n, overwrite, it)
