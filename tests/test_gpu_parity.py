"""GPU tier (-m gpu): the HIP path, called through the C ABI, against the CPU oracle on the same
seeded inputs.  fp results of SpMV and TRSV must be BIT-IDENTICAL to the reference order the oracle
restates (GPU fma == x86 fma); where the schedule legitimately differs (long rows in auto mode,
transposed SpMV, csrmm row-major) the componentwise forward-error bound of SURVEY.md section 8d is
asserted with the constant written here."""
import ctypes
import os

import numpy as np
import pytest

import oracle
from util import (EPS32, EPS64, abs_row_sums, banded_rows, kt_lanes, laplace5, pkg, powerlaw_rows, random_csr,
                  triangular_system, trsv_schedule)

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
P = pkg()
L = P.lib()


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    assert torch.cuda.is_available(), "GPU tests need a GPU (no CPU fallback exists)"
    st, d, cus, name = P.device_info()
    assert st == 0 and cus > 0
    yield
    L.aoclsparse_mi355_set_pointer_mode(P.PTR_AUTO)


def run_dmv(A, d, x, y0, alpha, beta, op=P.OP_NONE, on_device=True):
    if on_device:
        xd, yd = dev(x), dev(y0)
        st = P.dmv(op, alpha, A, d, xd, beta, yd)
        torch.cuda.synchronize()
        return st, yd.cpu().numpy()
    y = y0.copy()
    st = P.dmv(op, alpha, A, d, x, beta, y)
    return st, y


# --------------------------------------------------------------------------------------------------
# SpMV
# --------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("base", [0, 1])
@pytest.mark.parametrize("alpha,beta", [(1.0, 0.0), (5.1, 3.2), (-0.3, 1.0)])
def test_l100_dmv_bit_exact(base, alpha, beta):
    """BASELINE config 1/2: 10k x 10k 5-pt Laplacian; the reference runs ref_csrmv_gn (nnz <= 10 m)."""
    m, rp, ci, v = laplace5(100, base=base)
    v = v * np.random.default_rng(3).uniform(0.5, 1.5, len(v))  # inexact products
    x = np.sin(0.01 * np.arange(m))
    y0 = np.random.default_rng(4).uniform(-1, 1, m)
    A = P.Matrix(base, m, m, rp, ci, v)
    d = P.Descr(base=base)
    assert L.aoclsparse_set_mv_hint(A.h, P.OP_NONE, d.h, 10) == 0 and L.aoclsparse_optimize(A.h) == 0
    info = A.spmv_info()
    assert info.kernel in (3, 4) and info.order == 0 and info.device_resident == 1 and info.long_rows == 0  # SELL-64
    for on_device in (True, False):
        st, y = run_dmv(A, d, x, y0, alpha, beta, on_device=on_device)
        assert st == 0
        so, yr = oracle.dcsrmv(-1, base, alpha, m, len(v), v, ci, rp, x, beta, y0)
        assert so == 0 and np.array_equal(y, yr)


def test_beta_zero_overwrites_nan_y():
    m, rp, ci, v = laplace5(30)
    A = P.Matrix(0, m, m, rp, ci, v)
    d = P.Descr()
    x = np.ones(m)
    st, y = run_dmv(A, d, x, np.full(m, np.nan), 1.0, 0.0)
    so, yr = oracle.dcsrmv(-1, 0, 1.0, m, len(v), v, ci, rp, x, 0.0, np.zeros(m))
    assert st == 0 and np.array_equal(y, yr)


@pytest.mark.parametrize("kid,order", [(0, "ref"), (1, "lane4"), (2, "lane4"), (3, "lane8"), (-1, "lane8")])
def test_kid_selects_reference_order_bit_exact(kid, order):
    """nnz > 10 m: kid 0/1/2/3 = ref / AVX2 / AVX2 / AVX-512 summation orders (csrmv.hpp:338-343)."""
    m = n = 3000
    rp, ci, v = random_csr(11, m, n, lambda r, i: r.integers(0, 90))
    assert len(v) > 10 * m
    x = np.random.default_rng(5).uniform(-1, 1, n)
    y0 = np.random.default_rng(6).uniform(-1, 1, m)
    A = P.Matrix(0, m, n, rp, ci, v)
    d = P.Descr()
    if kid >= 0:
        assert L.aoclsparse_set_mv_hint_kid(A.h, P.OP_NONE, d.h, 0, kid) == 0
    st, y = run_dmv(A, d, x, y0, 2.5, -0.5)
    assert st == 0
    so, yr = oracle.dcsrmv_order(order, 0, 2.5, m, v, ci, rp, x, -0.5, y0)
    assert np.array_equal(y, yr), np.max(np.abs(y - yr))
    so, ya = oracle.dcsrmv(kid, 0, 2.5, m, len(v), v, ci, rp, x, -0.5, y0)
    assert np.array_equal(y, ya)


def test_invalid_kid():
    rp, ci, v = random_csr(12, 50, 50, lambda r, i: 20)
    A = P.Matrix(0, 50, 50, rp, ci, v)
    d = P.Descr()
    assert L.aoclsparse_set_mv_hint_kid(A.h, P.OP_NONE, d.h, 1, 7) == 0
    st, _ = run_dmv(A, d, np.ones(50), np.zeros(50), 1.0, 0.0)
    assert st == 14


def test_power_law_rows_with_long_rows_strict_and_auto():
    """scircuit / webbase-like: mean ~6, a few rows far longer than one LDS tile (2048)."""
    m = n = 40000
    rp, ci, v = random_csr(21, m, n, lambda r, i: 9000 if i in (17, 20011) else (
        2500 if i == 33333 else powerlaw_rows(6, 400)(r, i)))
    assert len(v) <= 10 * m
    x = np.random.default_rng(7).uniform(-1, 1, n)
    y0 = np.zeros(m)
    d = P.Descr()
    # auto: rows of fewer than tree_min (32) entries are bit-exact, longer rows within the componentwise bound
    # (the row-block kernel is what this test is about: with rows of 9,000 entries the automatic choice is merge-path since round 5)
    A = P.Matrix(0, m, n, rp, ci, v)
    assert L.aoclsparse_mi355_set_option(P.OPTION_SPMV_KERNEL, 1) == 0
    try:
        st, y = run_dmv(A, d, x, y0, 1.0, 0.0)
    finally:
        assert L.aoclsparse_mi355_set_option(P.OPTION_SPMV_KERNEL, 0) == 0
    assert st == 0 and A.spmv_info().long_rows == 3 and A.spmv_info().tree_min == 32
    so, yr = oracle.dcsrmv(-1, 0, 1.0, m, len(v), v, ci, rp, x, 0.0, y0)
    lens = np.diff(rp)
    short = lens < A.spmv_info().tree_min
    assert short.sum() > 0.9 * m and (~short).sum() > 50
    assert np.array_equal(y[short], yr[short])
    scale = abs_row_sums(rp, ci, v, x)
    c = 2 * np.ceil(np.log2(np.maximum(lens, 2))) + 4 + lens / 256.0  # tree + 256 partial chains
    assert np.all(np.abs(y - yr) <= c * EPS64 * scale + 1e-300)
    # pinned kernel (kid 0): strict reference order for every row, long ones included
    B = P.Matrix(0, m, n, rp, ci, v)
    assert L.aoclsparse_set_mv_hint_kid(B.h, P.OP_NONE, d.h, 0, 0) == 0
    st, ys = run_dmv(B, d, x, y0, 1.0, 0.0)
    assert st == 0 and np.array_equal(ys, yr) and B.spmv_info().tree_min == 0
    # ... and so does the strict option on the un-pinned handle
    assert L.aoclsparse_mi355_set_option(P.OPTION_SPMV_STRICT, 1) == 0
    try:
        st, ys = run_dmv(A, d, x, y0, 1.0, 0.0)
        assert st == 0 and np.array_equal(ys, yr)
    finally:
        assert L.aoclsparse_mi355_set_option(P.OPTION_SPMV_STRICT, 0) == 0


def _merge_cut_rows(rp, base, items=1024):
    """rows cut by a merge-path tile boundary (tiles of `items` row-ends + non-zeros)."""
    m = len(rp) - 1
    end = rp[1:].astype(np.int64) - base
    key = end + np.arange(m)  # strictly increasing: merge-path position of row end i
    d = np.arange(0, m + end[-1] + items, items)
    d = d[d <= m + end[-1]]
    i = np.searchsorted(key, d, side="left")
    j = d - i
    ok = i < m
    cut = np.zeros(m, dtype=bool)
    ii = i[ok]
    cut[ii[(rp[ii].astype(np.int64) - base) < j[ok]]] = True
    return cut


@pytest.fixture
def merge_kernel():
    assert L.aoclsparse_mi355_set_option(P.OPTION_SPMV_KERNEL, 2) == 0  # merge-path whenever it can serve the request
    yield
    assert L.aoclsparse_mi355_set_option(P.OPTION_SPMV_KERNEL, 0) == 0


@pytest.mark.parametrize("base", [0, 1])
def test_merge_path_kernel_power_law(merge_kernel, base):
    """merge-path tiles (aoclsparse_mi355_set_option(spmv_kernel, 2)): rows of fewer than tree_min (32) entries inside one tile
    follow the reference's scalar order bit for bit; a longer row, and a row cut by tile boundaries (the fixed-order sum of its
    pieces, combined by the look-back of the ONE launch, round 5), stay within (pieces + len) * eps * sum|a||x|."""
    m = n = 30000
    rp, ci, v = random_csr(121, m, n, lambda r, i: 9000 if i in (0, 17, 20011) else (
        2500 if i == m - 1 else (0 if i % 7 == 3 else powerlaw_rows(6, 400)(r, i))), base=base)
    assert len(v) <= 10 * m
    x = np.random.default_rng(7).uniform(-1, 1, n)
    y0 = np.random.default_rng(8).uniform(-1, 1, m)
    d = P.Descr(base=base)
    A = P.Matrix(base, m, n, rp, ci, v)
    for alpha, beta in ((1.0, 0.0), (-0.75, 1.5)):
        st, y = run_dmv(A, d, x, y0, alpha, beta)
        assert st == 0 and A.spmv_info().kernel == 2
        so, yr = oracle.dcsrmv(-1, base, alpha, m, len(v), v, ci, rp, x, beta, y0)
        cut = _merge_cut_rows(rp, base)
        assert 3 <= cut.sum() < 200
        lens = np.diff(rp)
        assert A.spmv_info().tree_min == 32
        exact = ~cut & (lens < 32)
        assert exact.sum() > 0.9 * m and np.array_equal(y[exact], yr[exact])
        scale = abs(alpha) * abs_row_sums(rp, ci, v, x, base) + abs(beta * y0)
        c = lens + lens / 1024.0 + 6
        assert np.all(np.abs(y - yr) <= c * EPS64 * scale + 1e-300)
    # a pinned kid keeps the strict order: served by the row-block kernel, cut rows included
    B = P.Matrix(base, m, n, rp, ci, v)
    assert L.aoclsparse_set_mv_hint_kid(B.h, P.OP_NONE, d.h, 0, 0) == 0
    st, ys = run_dmv(B, d, x, y0, 1.0, 0.0)
    so, yr = oracle.dcsrmv(-1, base, 1.0, m, len(v), v, ci, rp, x, 0.0, y0)
    assert st == 0 and np.array_equal(ys, yr)


def test_merge_path_kernel_on_many_streams_and_after_a_stream_is_destroyed(merge_kernel):
    """The head pieces of a launch travel through a granule set owned by the STREAM the launch runs on (round 5): launches of
    one handle interleaved on two live streams (different x) do not see each other's pieces; a stream created and destroyed
    through the HIP runtime is never touched again (the next launch on another stream succeeds); twenty streams in turn (more
    than the sets a plan keeps) still return the first launch's bits."""
    # streams are objects of ONE HIP runtime: this process may hold two copies (torch ships its own); take the one the library
    # is bound to
    st, path = P.hip_runtime_path()
    assert st == 0 and "libamdhip64" in path
    hip = ctypes.CDLL(path)
    hip.hipStreamCreate.argtypes, hip.hipStreamDestroy.argtypes = [ctypes.POINTER(ctypes.c_void_p)], [ctypes.c_void_p]
    hip.hipStreamSynchronize.argtypes = [ctypes.c_void_p]
    m = n = 20000
    rp, ci, v = random_csr(321, m, n, lambda r, i: 7000 if i in (5, 9000) else r.integers(0, 8))
    d = P.Descr()
    A = P.Matrix(0, m, n, rp, ci, v)
    xs = [np.random.default_rng(40 + k).uniform(-1, 1, n) for k in range(2)]
    xd = [dev(x) for x in xs]
    ref = []
    for k in range(2):
        yd = dev(np.zeros(m))
        assert P.dmv(P.OP_NONE, 1.0, A, d, xd[k], 0.0, yd) == 0
        torch.cuda.synchronize()
        ref.append(yd.cpu().numpy())
        so, yr = oracle.dcsrmv(-1, 0, 1.0, m, len(v), v, ci, rp, xs[k], 0.0, np.zeros(m))
        scale = abs_row_sums(rp, ci, v, xs[k])
        assert np.all(np.abs(ref[k] - yr) <= (np.diff(rp) + 24) * EPS64 * scale + 1e-300)
    assert A.spmv_info().kernel == 2
    L.aoclsparse_mi355_set_pointer_mode(P.PTR_DEVICE)
    streams = []
    try:
        for _ in range(20):
            h = ctypes.c_void_p()
            assert hip.hipStreamCreate(ctypes.byref(h)) == 0
            streams.append(h)
        ys = [dev(np.zeros(m)) for _ in range(2)]
        torch.cuda.synchronize()
        # two live streams, 30 launches each, interleaved, no synchronisation in between
        for it in range(30):
            for k in range(2):
                assert L.aoclsparse_mi355_set_stream(streams[k]) == 0
                assert P.dmv(P.OP_NONE, 1.0, A, d, xd[k], 0.0, ys[k]) == 0
        for k in range(2):
            assert hip.hipStreamSynchronize(streams[k]) == 0
            assert np.array_equal(ys[k].cpu().numpy(), ref[k])
        # stream 0 is destroyed; the plan's set for it is simply never used again
        assert L.aoclsparse_mi355_set_stream(streams[1]) == 0
        assert hip.hipStreamDestroy(streams[0]) == 0
        streams[0] = None
        ys[1].zero_()
        torch.cuda.synchronize()
        assert P.dmv(P.OP_NONE, 1.0, A, d, xd[1], 0.0, ys[1]) == 0
        assert hip.hipStreamSynchronize(streams[1]) == 0 and np.array_equal(ys[1].cpu().numpy(), ref[1])
        # more streams than a plan keeps granule sets for
        for h in streams[2:]:
            assert L.aoclsparse_mi355_set_stream(h) == 0
            ys[0].zero_()
            torch.cuda.synchronize()
            assert P.dmv(P.OP_NONE, 1.0, A, d, xd[0], 0.0, ys[0]) == 0
            assert hip.hipStreamSynchronize(h) == 0 and np.array_equal(ys[0].cpu().numpy(), ref[0])
    finally:
        assert L.aoclsparse_mi355_set_stream(None) == 0
        L.aoclsparse_mi355_set_pointer_mode(P.PTR_AUTO)
        torch.cuda.synchronize()
        for h in streams:
            if h is not None:
                hip.hipStreamDestroy(h)


def test_merge_path_kernel_edges(merge_kernel):
    d = P.Descr()
    # smaller than one tile; trailing and leading empty rows; one row spanning many tiles; rectangular
    for seed, m, n, rl in ((1, 40, 60, lambda r, i: r.integers(0, 9)),
                           (2, 5000, 300, lambda r, i: 0 if (i < 1100 or i > 3900) else r.integers(0, 12)),
                           (3, 3000, 20000, lambda r, i: 15000 if i == 1500 else 1),
                           (4, 1024, 1024, lambda r, i: 0)):
        rp, ci, v = random_csr(seed, m, n, rl)
        if len(v) == 0:
            ci, v = np.zeros(1, np.int32), np.zeros(1)
        x = np.random.default_rng(3).uniform(-1, 1, n)
        y0 = np.random.default_rng(4).uniform(-1, 1, m)
        A = P.Matrix(0, m, n, rp, ci, v)
        st, y = run_dmv(A, d, x, y0, 2.0, 0.5)
        so, yr = oracle.dcsrmv(-1, 0, 2.0, m, int(rp[-1]), v, ci, rp, x, 0.5, y0)
        assert st == 0
        cut = _merge_cut_rows(rp, 0) if rp[-1] > 0 else np.zeros(m, bool)
        exact = ~cut & (np.diff(rp) < 32)
        assert np.array_equal(y[exact], yr[exact])
        scale = 2.0 * abs_row_sums(rp, ci, v, x) + abs(0.5 * y0)
        assert np.all(np.abs(y - yr) <= (np.diff(rp) + 24) * EPS64 * scale + 1e-300)
    # a wide matrix (nnz > 10 m) runs the lane orders: the row-block kernel serves it, bit-exact
    rp, ci, v = random_csr(5, 2000, 2000, lambda r, i: r.integers(20, 40))
    x = np.random.default_rng(3).uniform(-1, 1, 2000)
    A = P.Matrix(0, 2000, 2000, rp, ci, v)
    st, y = run_dmv(A, d, x, np.zeros(2000), 1.0, 0.0)
    so, yr = oracle.dcsrmv(-1, 0, 1.0, 2000, len(v), v, ci, rp, x, 0.0, np.zeros(2000))
    assert st == 0 and np.array_equal(y, yr)


def test_unhinted_handle_is_promoted_to_sell():
    """no mv hint, no optimize: the first products run the row-block kernel; at the 8th the handle gets the SELL-64 copy
    an optimize would have built -- every product has the same bits; memory_usage_minimal keeps the handle as it is."""
    m, rp, ci, v = laplace5(300)
    x = np.random.default_rng(5).uniform(-1, 1, m)
    so, yr = oracle.dcsrmv(-1, 0, 1.0, m, len(v), v, ci, rp, x, 0.0, np.zeros(m))
    d = P.Descr()
    A = P.Matrix(0, m, m, rp, ci, v)
    kernels = []
    for _ in range(10):
        st, y = run_dmv(A, d, x, np.zeros(m), 1.0, 0.0)
        assert st == 0 and np.array_equal(y, yr)
        kernels.append(A.spmv_info().kernel)
    assert kernels[:7] == [1] * 7 and kernels[7:] == [4] * 3  # SELL-64; a stencil shares its (shifted) column lists
    Bm = P.Matrix(0, m, m, rp, ci, v)
    assert L.aoclsparse_set_memory_hint(Bm.h, 0) == 0  # aoclsparse_memory_usage_minimal
    for _ in range(10):
        st, y = run_dmv(Bm, d, x, np.zeros(m), 1.0, 0.0)
        assert st == 0 and np.array_equal(y, yr)
    assert Bm.spmv_info().kernel == 1


@pytest.mark.parametrize("kid,order", [(3, "lane8"), (1, "lane4")])
def test_long_rows_strict_lane_orders(kid, order):
    m, n = 300, 60000
    rp, ci, v = random_csr(22, m, n, lambda r, i: 10000 + i if i % 100 == 7 else r.integers(20, 60))
    x = np.random.default_rng(8).uniform(-1, 1, n)
    A = P.Matrix(0, m, n, rp, ci, v)
    d = P.Descr()
    assert L.aoclsparse_set_mv_hint_kid(A.h, P.OP_NONE, d.h, 0, kid) == 0
    st, y = run_dmv(A, d, x, np.zeros(m), 1.0, 0.0)
    so, yr = oracle.dcsrmv_order(order, 0, 1.0, m, v, ci, rp, x, 0.0, np.zeros(m))
    assert st == 0 and np.array_equal(y, yr)


def test_empty_rows_rectangular_and_unsorted_columns():
    m, n = 5000, 777
    rp, ci, v = random_csr(23, m, n, lambda r, i: 0 if i % 3 == 0 else r.integers(1, 25), base=1, sort=False)
    x = np.random.default_rng(9).uniform(-1, 1, n)
    y0 = np.random.default_rng(10).uniform(-1, 1, m)
    A = P.Matrix(1, m, n, rp, ci, v)
    d = P.Descr(base=1)
    st, y = run_dmv(A, d, x, y0, 0.7, 0.0)
    so, yr = oracle.dcsrmv(-1, 1, 0.7, m, len(v), v, ci, rp, x, 0.0, y0)
    assert st == 0 and np.array_equal(y, yr)
    assert np.all(y[::3] == 0.0)


def test_empty_matrix_scales_y():
    # mv.cpp:116-121
    A = P.Matrix(0, 4, 4, [0, 0, 0, 0, 0], np.zeros(1, np.int32), np.zeros(1))
    A2 = P.Matrix(0, 4, 4, np.zeros(5, np.int32), np.zeros(0, np.int32), np.zeros(0))
    d = P.Descr()
    st, y = run_dmv(A2, d, np.ones(4), np.arange(4.0), 3.0, 2.0)
    assert st == 0 and np.array_equal(y, 2.0 * np.arange(4.0))
    st, y = run_dmv(A2, d, np.ones(4), np.full(4, np.nan), 3.0, 0.0, on_device=False)
    assert st == 0 and np.array_equal(y, np.zeros(4))


def test_float_smv_bit_exact():
    """float general SpMV always runs the 8-lane AVX2 kernel (csrmv.hpp:317-321)."""
    for seed, rl in ((31, lambda r, i: r.integers(0, 70)), (32, lambda r, i: r.integers(0, 7))):
        m = n = 2500
        rp, ci, v = random_csr(seed, m, n, rl, dtype=np.float32)
        x = np.random.default_rng(1).uniform(-1, 1, n).astype(np.float32)
        y0 = np.random.default_rng(2).uniform(-1, 1, m).astype(np.float32)
        A = P.Matrix(0, m, n, rp, ci, v)
        d = P.Descr()
        xd, yd = dev(x), dev(y0)
        st = P.smv(P.OP_NONE, 1.5, A, d, xd, 0.25, yd)
        torch.cuda.synchronize()
        so, yr = oracle.scsrmv("lane8", 0, 1.5, m, v, ci, rp, x, 0.25, y0)
        assert st == 0 and np.array_equal(yd.cpu().numpy(), yr)
        y = y0.copy()
        st = P.scsrmv(P.OP_NONE, 1.5, m, n, len(v), v, ci, rp, d, x, 0.25, y)
        assert st == 0 and np.array_equal(y, yr)


def test_raw_dcsrmv_host_and_device_pointers():
    m, rp, ci, v = laplace5(64)
    x = np.random.default_rng(1).uniform(-1, 1, m)
    y0 = np.random.default_rng(2).uniform(-1, 1, m)
    d = P.Descr()
    so, yr = oracle.dcsrmv(-1, 0, 1.0, m, len(v), v, ci, rp, x, 0.5, y0)
    y = y0.copy()
    assert P.dcsrmv(P.OP_NONE, 1.0, m, m, len(v), v, ci, rp, d, x, 0.5, y) == 0
    assert np.array_equal(y, yr)
    vd, cd, rd, xd, yd = dev(v), dev(ci), dev(rp), dev(x), dev(y0)
    for _ in range(2):  # second call hits the cached row-block plan
        yd.copy_(torch.from_numpy(y0))
        assert P.dcsrmv(P.OP_NONE, 1.0, m, m, len(v), vd, cd, rd, d, xd, 0.5, yd) == 0
        torch.cuda.synchronize()
        assert np.array_equal(yd.cpu().numpy(), yr)
    # thin HIP C-ABI with an explicit host-built plan
    for tile in (512, 1024, 2048):
        rb = np.zeros(L.mi355_csrmv_plan_bound(m, len(v)), dtype=np.int32)
        nb = L.mi355_csrmv_plan_host(m, 0, tile, P._ptr(rp), P._ptr(rb))
        rbd = dev(rb)
        yd.copy_(torch.from_numpy(y0))
        st = L.mi355_dcsrmv(None, 0, 0, tile, 0, 1.0, m, P._ptr(vd), P._ptr(cd), P._ptr(rd), P._ptr(rbd), nb,
                            P._ptr(xd), 0.5, P._ptr(yd))
        torch.cuda.synchronize()
        assert st == 0 and np.array_equal(yd.cpu().numpy(), yr)
    # arrays that are only 8-byte aligned take the scalar-load path of the kernel
    vo, co = dev(np.concatenate([[0.0], v])), dev(np.concatenate([[0], ci]).astype(np.int32))
    yd.copy_(torch.from_numpy(y0))
    st = L.mi355_dcsrmv(None, 0, 0, 2048, 0, 1.0, m, vo.data_ptr() + 8, co.data_ptr() + 4, P._ptr(rd),
                        P._ptr(rbd), nb, P._ptr(xd), 0.5, P._ptr(yd))
    torch.cuda.synchronize()
    assert st == 0 and np.array_equal(yd.cpu().numpy(), yr)


@pytest.mark.parametrize("op", [P.OP_TRANSPOSE, P.OP_CONJ_TRANSPOSE])
def test_transposed_spmv_within_bound(op):
    """csrmvt_kt's result depends on the CPU thread count (per-thread buffers); parity is the
    componentwise bound |dy| <= (len+4) eps sum|a x||alpha| + 2 eps |beta y| against the 1-thread order."""
    m, n = 4000, 3000
    rp, ci, v = random_csr(41, m, n, lambda r, i: r.integers(0, 30))
    x = np.random.default_rng(3).uniform(-1, 1, m)
    y0 = np.random.default_rng(4).uniform(-1, 1, n)
    A = P.Matrix(0, m, n, rp, ci, v)
    d = P.Descr()
    assert L.aoclsparse_set_mv_hint(A.h, op, d.h, 5) == 0 and L.aoclsparse_optimize(A.h) == 0
    assert A.spmv_info(op).device_resident == 1
    st, y = run_dmv(A, d, x, y0, 5.1, 3.2, op=op)
    so, yr = oracle.dcsrmvt(0, 5.1, m, n, v, ci, rp, x, 3.2, y0)
    assert st == 0 and so == 0
    st2, cp, ri, cv = oracle.dcsr2csc(m, n, len(v), 0, 0, rp, ci, v)
    scale = abs_row_sums(cp, ri, cv, x) * 5.1
    lens = np.diff(cp)
    assert np.all(np.abs(y - yr) <= (lens + 4) * EPS64 * scale + 2 * EPS64 * np.abs(3.2 * y0) + 1e-300)
    yh = y0.copy()
    assert P.dcsrmv(op, 5.1, m, n, len(v), v, ci, rp, d, x, 3.2, yh) == 0
    assert np.array_equal(yh, y)


def test_large_laplacian_linearity_and_checksum():
    """Full-size property check (grid 2048^2, 21 M nnz): A*1 has a known closed form and SpMV is linear."""
    g = 2048
    m, rp, ci, v = laplace5(g)
    A = P.Matrix(0, m, m, rp, ci, v)
    d = P.Descr()
    assert L.aoclsparse_set_mv_hint(A.h, P.OP_NONE, d.h, 100) == 0 and L.aoclsparse_optimize(A.h) == 0
    ones = torch.ones(m, dtype=torch.float64, device="cuda")
    y = torch.empty_like(ones)
    assert P.dmv(P.OP_NONE, 1.0, A, d, ones, 0.0, y) == 0
    torch.cuda.synchronize()
    deg = np.diff(rp) - 1
    assert np.array_equal(y.cpu().numpy(), 4.0 - deg)  # exact in fp64
    x1 = torch.rand(m, dtype=torch.float64, device="cuda")
    y1, y2 = torch.empty_like(x1), torch.empty_like(x1)
    assert P.dmv(P.OP_NONE, 1.0, A, d, x1, 0.0, y1) == 0
    assert P.dmv(P.OP_NONE, 1.0, A, d, 2.0 * x1, 0.0, y2) == 0  # scaling by 2 is exact
    torch.cuda.synchronize()
    assert torch.equal(y2, 2.0 * y1)
    # spot-check 100k rows against the oracle
    xs = x1.cpu().numpy()
    so, yr = oracle.dcsrmv(-1, 0, 1.0, m, len(v), v, ci, rp, xs, 0.0, np.zeros(m))
    assert np.array_equal(y1.cpu().numpy(), yr)


# --------------------------------------------------------------------------------------------------
# TRSV
# --------------------------------------------------------------------------------------------------
KIND = {("lower", "n"): "l", ("lower", "t"): "lt", ("upper", "n"): "u", ("upper", "t"): "ut"}


def oracle_trsv(base, m, rp, ci, v, fill, trans, unit, alpha, b, kid=None):
    """the serial chain the reference runs for this kid: ref_trsv_* for kid 0 / auto (what every GPU schedule reproduces),
    kt_trsv_* with 4 / 8 lanes for kid 1, 2 / 3 (trsv.cpp:321-353; the transposed KT kernels equal the reference's bits)"""
    o = oracle.dcsr_optimize(m, m, len(v), base, rp, ci, v)
    assert o["status"] == 0
    ilend = o["idiag"] if fill == "lower" else o["iurow"]
    lanes = kt_lanes(kid)
    if lanes:
        st, x = oracle.trsv_kt(KIND[(fill, trans)], lanes, alpha, m, o["base"], o["val"], o["ind"], o["ptr"], ilend, b, unit)
    else:
        st, x = oracle.dtrsv(KIND[(fill, trans)], alpha, m, o["base"], o["val"], o["ind"], o["ptr"], ilend, b, unit)
    assert st == 0
    return x


@pytest.mark.parametrize("kid", [None, 0, 1, 3])
def test_trsv_reference_kats(kats, kid):
    """D7 / S7 / N25 systems of the reference's unit tests (trsv_tests.cpp:279-318), every kid (= the arithmetic of the
    kernel the reference dispatches for it), both index bases, triangular and symmetric descriptors, host and device vectors."""
    tol = kats["trsv_abs_tol"]
    for c in kats["trsv"]:
        for base in (0, 1):
            m = c["m"]
            rp = np.array(c["row_ptr"], np.int32) + base
            ci = np.array(c["col_ind"], np.int32) + base
            v = np.array(c["val"], np.float64)
            b = np.array(c["b"], np.float64)
            A = P.Matrix(base, m, m, rp, ci, v)
            mt = P.TYPE_TRIANGULAR if base == 0 else P.TYPE_SYMMETRIC
            d = P.Descr(base=base, mtype=mt, fill=P.FILL_LOWER if c["fill"] == "lower" else P.FILL_UPPER,
                        diag=P.DIAG_UNIT if c["unit"] else P.DIAG_NON_UNIT)
            op = P.OP_NONE if c["trans"] == "n" else P.OP_TRANSPOSE
            x = np.zeros(m)
            st = P.dtrsv(op, c["alpha"], A, d, b, x, kid=kid)
            assert st == 0, (c["name"], base, kid, P.STATUS[st])
            assert np.max(np.abs(x - np.array(c["xref"]))) <= tol, (c["name"], base, kid)
            xr = oracle_trsv(base, m, rp, ci, v, c["fill"], c["trans"], c["unit"], c["alpha"], b, kid=kid)
            assert np.array_equal(x, xr), (c["name"], base, kid)
            bd, xd = dev(b), torch.zeros(m, dtype=torch.float64, device="cuda")
            assert P.dtrsv(op, c["alpha"], A, d, bd, xd, kid=kid) == 0
            torch.cuda.synchronize()
            assert np.array_equal(xd.cpu().numpy(), xr)


@pytest.mark.parametrize("fill,trans,unit", [("lower", "n", False), ("lower", "n", True), ("lower", "t", False),
                                             ("upper", "n", False), ("upper", "t", True), ("upper", "t", False)])
@pytest.mark.parametrize("kid", [0, 2, 3])
def test_trsv_random_bit_exact(fill, trans, unit, kid):
    """kid 0: ref_trsv_*; kid 2: the 256-bit KT kernel's order; kid 3: the 512-bit one -- bit for bit (rows of up to 12 entries:
    full vector groups, masked remainders and scalar tails all occur)"""
    m = 20000
    rp, ci, v = triangular_system(51, m, 6, band=300)
    b = np.random.default_rng(5).uniform(-1, 1, m)
    A = P.Matrix(0, m, m, rp, ci, v)
    d = P.Descr(mtype=P.TYPE_TRIANGULAR, fill=P.FILL_LOWER if fill == "lower" else P.FILL_UPPER,
                diag=P.DIAG_UNIT if unit else P.DIAG_NON_UNIT)
    op = P.OP_NONE if trans == "n" else P.OP_TRANSPOSE
    assert L.aoclsparse_set_sv_hint(A.h, op, d.h, 3) == 0 and L.aoclsparse_optimize(A.h) == 0
    lv = A.trsv_levels(P.FILL_LOWER if fill == "lower" else P.FILL_UPPER, op)
    assert 1 < lv < m
    bd, xd = dev(b), torch.zeros(m, dtype=torch.float64, device="cuda")
    st = P.dtrsv(op, 1.3, A, d, bd, xd, kid=kid)
    torch.cuda.synchronize()
    xr = oracle_trsv(0, m, rp, ci, v, fill, trans, unit, 1.3, b, kid=kid)
    assert st == 0 and np.array_equal(xd.cpu().numpy(), xr)
    if kid == 0:
        # every schedule returns the chain's bits (the kid picks the arithmetic, the schedule only who computes what when)
        for sched in (0, 1, 2, 3, 4):
            with trsv_schedule(P, sched):
                xd.zero_()
                assert P.dtrsv(op, 1.3, A, d, bd, xd) == 0
                torch.cuda.synchronize()
                assert np.array_equal(xd.cpu().numpy(), xr), sched
    elif trans == "n":
        # the KT order is served by the per-level launches and by the lane-per-position sync-free kernel
        for sched in (0, 2):
            with trsv_schedule(P, sched):
                xd.zero_()
                assert P.dtrsv(op, 1.3, A, d, bd, xd, kid=kid) == 0
                torch.cuda.synchronize()
                assert np.array_equal(xd.cpu().numpy(), xr), sched
        assert not np.array_equal(xr, oracle_trsv(0, m, rp, ci, v, fill, trans, unit, 1.3, b))  # a different summation


def test_trsv_unsorted_input_with_missing_diagonals_unit():
    """optimize must sort + insert zero diagonals (hint_tests.cpp N5_1_hole); unit-diag solve works."""
    m = 3000
    rp, ci, v = random_csr(61, m, m, lambda r, i: r.integers(0, 9), sort=False)
    b = np.random.default_rng(6).uniform(-1, 1, m)
    A = P.Matrix(0, m, m, rp, ci, v)
    d = P.Descr(mtype=P.TYPE_TRIANGULAR, fill=P.FILL_LOWER, diag=P.DIAG_UNIT)
    x = np.zeros(m)
    assert P.dtrsv(P.OP_NONE, 1.0, A, d, b, x) == 0
    xr = oracle_trsv(0, m, rp, ci, v, "lower", "n", True, 1.0, b)
    assert np.array_equal(x, xr)
    dn = P.Descr(mtype=P.TYPE_TRIANGULAR, fill=P.FILL_LOWER, diag=P.DIAG_NON_UNIT)
    assert P.dtrsv(P.OP_NONE, 1.0, A, dn, b, x) == 5  # not full rank: invalid_value (trsv.cpp:133-137)


def test_trsv_strided_and_float():
    m = 4000
    rp, ci, v = triangular_system(71, m, 4, band=50)
    A = P.Matrix(0, m, m, rp, ci, v)
    d = P.Descr(mtype=P.TYPE_TRIANGULAR, fill=P.FILL_UPPER)
    rng = np.random.default_rng(7)
    incb, incx = 3, 2
    b = rng.uniform(-1, 1, m * incb)
    x = np.full(m * incx, 7.0)
    assert P.dtrsv(P.OP_NONE, 0.9, A, d, b, x, incb=incb, incx=incx) == 0
    xr = oracle_trsv(0, m, rp, ci, v, "upper", "n", False, 0.9, b[::incb][:m].copy())
    assert np.array_equal(x[::incx][:m], xr) and np.all(x[1::incx] == 7.0)
    bd, xd = dev(b), dev(np.full(m * incx, 7.0))
    assert P.dtrsv(P.OP_NONE, 0.9, A, d, bd, xd, incb=incb, incx=incx) == 0
    torch.cuda.synchronize()
    assert np.array_equal(xd.cpu().numpy(), x)
    # float: same recurrence in fp32 (oracle orc_strsv_l)
    vf = v.astype(np.float32)
    Af = P.Matrix(0, m, m, rp, ci, vf)
    dl = P.Descr(mtype=P.TYPE_TRIANGULAR, fill=P.FILL_LOWER)
    bf = rng.uniform(-1, 1, m).astype(np.float32)
    xf = np.zeros(m, np.float32)
    assert P.strsv(P.OP_NONE, 1.0, Af, dl, bf, xf) == 0
    o = oracle.dcsr_optimize(m, m, len(v), 0, rp, ci, v)
    xo = np.zeros(m, np.float32)
    st = oracle.lib().orc_strsv_l(ctypes.c_float(1.0), m, 0, P._ptr(vf), P._ptr(ci), P._ptr(rp),
                                  P._ptr(o["idiag"]), P._ptr(bf), 1, P._ptr(xo), 1, 0)
    assert st == 0 and np.array_equal(xf, xo)


def test_trsv_ilu0_laplacian_residual():
    """BASELINE config 5 in miniature: unit-lower ILU(0) factor of a 2-D Laplacian (grid 300^2):
    2g-1 dependency levels, b = L*1 so x* = 1; residual and bit-parity with the serial CPU solve."""
    g = 300
    m, rp, ci, v = laplace5(g)
    st, lu, diag = oracle.dilu0(m, 0, rp, ci, v)
    assert st == 0
    A = P.Matrix(0, m, m, rp, ci, lu)
    d = P.Descr(mtype=P.TYPE_TRIANGULAR, fill=P.FILL_LOWER, diag=P.DIAG_UNIT)
    assert L.aoclsparse_set_sv_hint(A.h, P.OP_NONE, d.h, 10) == 0 and L.aoclsparse_optimize(A.h) == 0
    assert A.trsv_levels(P.FILL_LOWER) == 2 * g - 1
    o = oracle.dcsr_optimize(m, m, len(lu), 0, rp, ci, lu)
    # b = L * 1 computed with the oracle's strict-lower rows
    b = np.ones(m)
    low = np.zeros(m)
    for i in range(m):
        low[i] = lu[rp[i]:o["idiag"][i]].sum()
    b = b + low
    for kid in (0, 1, None):
        xd = torch.zeros(m, dtype=torch.float64, device="cuda")
        assert P.dtrsv(P.OP_NONE, 1.0, A, d, dev(b), xd, kid=kid) == 0
        torch.cuda.synchronize()
        x = xd.cpu().numpy()
        st, xr = oracle.dtrsv("l", 1.0, m, 0, lu, ci, rp, o["idiag"], b, True)
        assert np.array_equal(x, xr)
        assert np.max(np.abs(x - 1.0)) < 1e-12


# --------------------------------------------------------------------------------------------------
# csrmm
# --------------------------------------------------------------------------------------------------
def test_csrmm_reference_kats(kats):
    for c in kats["csrmm"]:
        m, k, n = c["m"], c["k"], c["n"]
        A = P.Matrix(0, m, k, c["row_ptr"], c["col_ind"], np.array(c["val"], np.float64))
        d = P.Descr()
        for order, ldb, ldc, key in ((P.ORDER_COLUMN, k, m, "C_exp_col"), (P.ORDER_ROW, n, n, "C_exp_row")):
            B, C = np.array(c["B"], np.float64), np.array(c["C"], np.float64)
            assert P.dcsrmm(P.OP_NONE, c["alpha"], A, d, order, B, n, ldb, c["beta"], C, ldc) == 0
            assert np.allclose(C[: m * n], np.array(c[key]), rtol=1e-13, atol=1e-12), (c["name"], order)
    c = [t for t in kats["csrmm"] if t["name"] == "id1_5x5"][0]
    A = P.Matrix(0, 5, 5, c["row_ptr"], c["col_ind"], np.array(c["val"], np.float64))
    d = P.Descr()
    for order, key in ((P.ORDER_COLUMN, "C_exp_col_T"), (P.ORDER_ROW, "C_exp_row_T")):
        B, C = np.array(c["B"], np.float64), np.array(c["C"], np.float64)
        assert P.dcsrmm(P.OP_TRANSPOSE, c["alpha"], A, d, order, B, 5, 5, c["beta"], C, 5) == 0
        assert np.allclose(C, np.array(c[key]), rtol=1e-13, atol=1e-12)


@pytest.mark.parametrize("n", [1, 7, 32, 256])
@pytest.mark.parametrize("alpha,beta", [(1.0, 0.0), (3.0, -2.0)])
def test_csrmm_random_vs_oracle(n, alpha, beta):
    """Column-major: bit-identical to csrmm_col_major_ref's order.  Row-major: the same per-element
    arithmetic is used (dot then fma(beta,C,alpha*dot)), so it equals the col-major oracle bitwise and
    sits within (len+3) eps sum|a b||alpha| + 2 eps |beta c| of csrmm_row_major_ref."""
    m, k = 3000, 2500
    rp, ci, v = random_csr(81, m, k, lambda r, i: r.integers(0, 14), base=1)
    rng = np.random.default_rng(777)
    A = P.Matrix(1, m, k, rp, ci, v)
    d = P.Descr(base=1)
    assert L.aoclsparse_set_mm_hint(A.h, P.OP_NONE, d.h, 2) == 0 and L.aoclsparse_optimize(A.h) == 0
    ldb, ldc = k + 3, m + 5
    B = rng.uniform(-1, 1, ldb * n)
    C0 = rng.uniform(-1, 1, ldc * n)
    Cd = dev(C0)
    assert P.dcsrmm(P.OP_NONE, alpha, A, d, P.ORDER_COLUMN, dev(B), n, ldb, beta, Cd, ldc) == 0
    torch.cuda.synchronize()
    so, Cr = oracle.dcsrmm("col", alpha, 1, v, ci, rp, m, B, n, ldb, beta, C0, ldc)
    assert np.array_equal(Cd.cpu().numpy(), Cr)
    # row-major, host pointers
    ldb, ldc = n + 2, n + 4
    Br = rng.uniform(-1, 1, k * ldb)
    Cr0 = rng.uniform(-1, 1, m * ldc)
    Ch = Cr0.copy()
    assert P.dcsrmm(P.OP_NONE, alpha, A, d, P.ORDER_ROW, Br, n, ldb, beta, Ch, ldc) == 0
    so, Cref = oracle.dcsrmm("row", alpha, 1, v, ci, rp, m, Br, n, ldb, beta, Cr0, ldc)
    Bm, Cm = Br.reshape(k, ldb)[:, :n], Cr0.reshape(m, ldc)[:, :n]
    Av = np.abs(v)
    scale = np.zeros((m, n))
    for i in range(m):
        s, e = rp[i] - 1, rp[i + 1] - 1
        if e > s:
            scale[i] = Av[s:e] @ np.abs(Bm[ci[s:e] - 1])
    lens = np.diff(rp)[:, None]
    diff = np.abs(Ch.reshape(m, ldc)[:, :n] - Cref.reshape(m, ldc)[:, :n])
    assert np.all(diff <= (lens + 3) * EPS64 * scale * abs(alpha) + 2 * EPS64 * np.abs(beta * Cm) + 1e-300)
    assert np.array_equal(Ch.reshape(m, ldc)[:, n:], Cr0.reshape(m, ldc)[:, n:])  # padding untouched


@pytest.mark.parametrize("base", [0, 1])
def test_csrmm_row_groups_block_structured(base):
    """rows that share a column pattern (the dof rows of a node) are multiplied as one group per wavefront; per element
    the chain is still the row in CSR order, so C equals the column-major kernel's (= csrmm_col_major_ref's) bits.
    Groups of 1..11 rows (cap 8), a group cut by an empty row, 128 and 384 columns with padded leading dimensions."""
    rng = np.random.default_rng(5)
    nodes, k = 260, 1300
    rows_p, rows_c = [0], []
    for nd in range(nodes):
        dof = 1 + nd % 11
        cols = np.sort(rng.choice(k, size=3 + nd % 37, replace=False))
        for r in range(dof):
            if nd % 17 == 3 and r == 2:
                rows_p.append(rows_p[-1])  # an empty row inside the node
                continue
            rows_c.append(cols)
            rows_p.append(rows_p[-1] + len(cols))
    m = len(rows_p) - 1
    rp = np.array(rows_p, np.int32) + base
    ci = (np.concatenate(rows_c) + base).astype(np.int32)
    v = rng.uniform(-1, 1, len(ci))
    A = P.Matrix(base, m, k, rp, ci, v)
    d = P.Descr(base=base)
    for n, alpha, beta in ((128, 1.0, 0.0), (384, -0.7, 1.3), (32, 1.0, 0.0), (96, 2.0, -1.0)):
        ldb, ldc = n + 2, n + 6
        Br, C0 = rng.uniform(-1, 1, k * ldb), rng.uniform(-1, 1, m * ldc)
        Cd = dev(C0)
        assert P.dcsrmm(P.OP_NONE, alpha, A, d, P.ORDER_ROW, dev(Br), n, ldb, beta, Cd, ldc) == 0
        torch.cuda.synchronize()
        info = A.spmv_info()
        assert 0 < info.mm_groups < m / 1.5
        # the column-major oracle on the transposed layouts gives the per-element reference bits
        Bc = np.ascontiguousarray(Br.reshape(k, ldb)[:, :n].T).ravel()
        Cc = np.ascontiguousarray(C0.reshape(m, ldc)[:, :n].T).ravel()
        so, Cref = oracle.dcsrmm("col", alpha, base, v, ci, rp, m, Bc, n, k, beta, Cc, m)
        got = Cd.cpu().numpy().reshape(m, ldc)
        assert np.array_equal(got[:, :n], Cref.reshape(n, m).T)
        assert np.array_equal(got[:, n:], C0.reshape(m, ldc)[:, n:])
        # column-major operands with rows this long take the detour through the row-major kernels (relayout of B and
        # C): same bits as csrmm_col_major_ref, leading-dimension padding untouched
        ldb2, ldc2 = k + 5, m + 3
        B2, C2 = rng.uniform(-1, 1, ldb2 * n), rng.uniform(-1, 1, ldc2 * n)
        C2d = dev(C2)
        assert P.dcsrmm(P.OP_NONE, alpha, A, d, P.ORDER_COLUMN, dev(B2), n, ldb2, beta, C2d, ldc2) == 0
        torch.cuda.synchronize()
        so, C2ref = oracle.dcsrmm("col", alpha, base, v, ci, rp, m, B2, n, ldb2, beta, C2, ldc2)
        assert len(v) > 8 * m and np.array_equal(C2d.cpu().numpy(), C2ref)


@pytest.mark.parametrize("dof", [2, 3, 4, 5, 6, 8])
def test_csrmm_column_major_groups_write_c_in_place(dof):
    """column-major operands on a handle with row groups, n >= 128: only B goes through the row-major scratch, the row-group
    kernel writes column-major C itself (an LDS tile per workgroup, column by column).  Every group size the kernel is
    compiled for, a last workgroup with fewer than four groups, a partial last 128-column chunk, beta classes, padded
    leading dimensions untouched; bits of csrmm_col_major_ref."""
    rng = np.random.default_rng(40 + dof)
    nodes, k = 203, 900  # 203 groups: the last workgroup holds three
    rows_p, rows_c = [0], []
    for nd in range(nodes):
        cols = np.sort(rng.choice(k, size=12 + nd % 29, replace=False))
        for r in range(dof):
            rows_c.append(cols)
            rows_p.append(rows_p[-1] + len(cols))
    m = len(rows_p) - 1
    rp = np.array(rows_p, np.int32)
    ci = np.concatenate(rows_c).astype(np.int32)
    v = rng.uniform(-1, 1, len(ci))
    A = P.Matrix(0, m, k, rp, ci, v)
    d = P.Descr()
    # (32 <= n < 128: the sub-wave row-group kernels, a group per 16 / 32 lanes; n >= 128: a group per wavefront)
    for n, alpha, beta in ((128, 1.0, 0.0), (200, -0.5, 2.0), (256, 2.0, 0.0), (32, 1.0, 0.0), (46, 2.0, -1.0), (64, 1.0, 0.5), (100, -1.0, 0.0)):
        ldb, ldc = k + 3, m + 5
        B, C0 = rng.uniform(-1, 1, ldb * n), rng.uniform(-1, 1, ldc * n)
        Cd = dev(C0)
        assert P.dcsrmm(P.OP_NONE, alpha, A, d, P.ORDER_COLUMN, dev(B), n, ldb, beta, Cd, ldc) == 0
        torch.cuda.synchronize()
        assert A.spmv_info().mm_groups == nodes
        so, Cr = oracle.dcsrmm("col", alpha, 0, v, ci, rp, m, B, n, ldb, beta, C0, ldc)
        assert so == 0 and np.array_equal(Cd.cpu().numpy(), Cr), "dof=%d n=%d" % (dof, n)
    # float: the column-major result equals the row-major one of the same handle (same chains; that path has its own tests)
    Af = P.Matrix(0, m, k, rp, ci, v.astype(np.float32))
    n = 130
    Bm, C0 = rng.uniform(-1, 1, (k, n)).astype(np.float32), rng.uniform(-1, 1, (m, n)).astype(np.float32)
    Cr_, Cc_ = dev(C0.ravel()), dev(np.ascontiguousarray(C0.T).ravel())
    assert P.scsrmm(P.OP_NONE, 1.5, Af, d, P.ORDER_ROW, dev(Bm.ravel()), n, n, -0.5, Cr_, n) == 0
    assert P.scsrmm(P.OP_NONE, 1.5, Af, d, P.ORDER_COLUMN, dev(np.ascontiguousarray(Bm.T).ravel()), n, k, -0.5, Cc_, m) == 0
    torch.cuda.synchronize()
    assert np.array_equal(Cc_.cpu().numpy().reshape(n, m).T, Cr_.cpu().numpy().reshape(m, n))


def test_csrmm_alpha_zero_and_transpose():
    m, k, n = 500, 400, 16
    rp, ci, v = random_csr(91, m, k, lambda r, i: r.integers(0, 9))
    A = P.Matrix(0, m, k, rp, ci, v)
    d = P.Descr()
    rng = np.random.default_rng(1)
    C0 = rng.uniform(-1, 1, m * n)
    C = C0.copy()
    assert P.dcsrmm(P.OP_NONE, 0.0, A, d, P.ORDER_ROW, np.ones(k * n), n, n, 0.0, C, n) == 0
    assert np.all(C == 0.0)
    C = C0.copy()
    C[3] = np.nan
    assert P.dcsrmm(P.OP_NONE, 0.0, A, d, P.ORDER_ROW, np.ones(k * n), n, n, 0.0, C, n) == 0
    assert np.all(C == 0.0)  # scale_dense_matrix writes exact zeros (csrmm.hpp:366-393)
    # op = transpose: C (k x n) = A^T B (m x n)
    B = rng.uniform(-1, 1, m * n)
    Ct = rng.uniform(-1, 1, k * n)
    Ch = Ct.copy()
    assert P.dcsrmm(P.OP_TRANSPOSE, 2.0, A, d, P.ORDER_ROW, B, n, n, 0.5, Ch, n) == 0
    st, cp, ri, cv = oracle.dcsr2csc(m, k, len(v), 0, 0, rp, ci, v)
    so, Cref = oracle.dcsrmm("col", 2.0, 0, cv, ri, cp, k, np.ascontiguousarray(B.reshape(m, n).T).ravel(), n, m, 0.5,
                             np.ascontiguousarray(Ct.reshape(k, n).T).ravel(), k)
    assert np.array_equal(Ch.reshape(k, n), Cref.reshape(n, k).T)


def test_csrmm_column_shards_compose():
    """Multi-GPU contract: C[:, J] depends only on A and B[:, J]; shards are independent."""
    g = 200
    m, rp, ci, v = laplace5(g)
    n, shards = 64, 4
    rng = np.random.default_rng(777)
    B = rng.uniform(-1, 1, (n, m))  # column-major storage: B[j] is column j
    A = P.Matrix(0, m, m, rp, ci, v)
    d = P.Descr()
    full = torch.zeros(n * m, dtype=torch.float64, device="cuda")
    assert P.dcsrmm(P.OP_NONE, 1.0, A, d, P.ORDER_COLUMN, dev(B.ravel()), n, m, 0.0, full, m) == 0
    w = n // shards
    parts = []
    for s in range(shards):
        Cs = torch.zeros(w * m, dtype=torch.float64, device="cuda")
        assert P.dcsrmm(P.OP_NONE, 1.0, A, d, P.ORDER_COLUMN, dev(B[s * w:(s + 1) * w].ravel()), w, m, 0.0, Cs, m) == 0
        parts.append(Cs)
    torch.cuda.synchronize()
    assert torch.equal(torch.cat(parts), full)


# --------------------------------------------------------------------------------------------------
# symmetric / triangular descriptors (derived general CSR on the device, csrc/derived.cpp)
# --------------------------------------------------------------------------------------------------
def _dense(m, n, rp, ci, v, base=0):
    A = np.zeros((m, n))
    for i in range(m):
        A[i, ci[rp[i] - base:rp[i + 1] - base] - base] = v[rp[i] - base:rp[i + 1] - base]
    return A


def test_symmetric_raw_csrmv_kat(kats):
    """csrmv_tests.cpp:288-352: one stored triangle, symmetric descriptor, op = transpose, y = NaN, beta = 0."""
    c = kats["csrmv_sym"][0]
    m = c["m"]
    rp, ci, v = np.array(c["row_ptr"], np.int32), np.array(c["col_ind"], np.int32), np.array(c["val"], np.float64)
    x = np.array(c["x"], np.float64)
    d = P.Descr(mtype=P.TYPE_SYMMETRIC, fill=P.FILL_LOWER)
    for op in (P.OP_NONE, P.OP_TRANSPOSE):
        y = np.full(m, np.nan)
        assert P.dcsrmv(op, c["alpha"], m, m, len(v), v, ci, rp, d, x, c["beta"], y) == 0
        assert np.array_equal(y, np.array(c["y_gold"], np.float64))
    so, yo = oracle.dcsrmv_symm_raw(0, 1.0, m, v, ci, rp, x, 0.0, np.zeros(m))
    assert np.array_equal(yo, np.array(c["y_gold"], np.float64))


@pytest.mark.parametrize("fill", [P.FILL_LOWER, P.FILL_UPPER])
@pytest.mark.parametrize("diag", [P.DIAG_NON_UNIT, P.DIAG_UNIT, P.DIAG_ZERO])
def test_symmetric_and_triangular_dmv(fill, diag):
    """Against the serial reference kernels restated in the oracle (csrmv_kr.hpp:107-186, 577-728) within
    |dy| <= (len+6) eps (|alpha| sum|a x| + |beta y|); the operators are also checked against a dense
    construction so that the expansion itself (which triangle, which diagonal) is pinned exactly."""
    m = 1500
    rp, ci, v = random_csr(101 + fill + 2 * diag, m, m, lambda r, i: r.integers(0, 16), sort=False)
    A = P.Matrix(0, m, m, rp, ci, v)
    rng = np.random.default_rng(9)
    x, y0 = rng.uniform(-1, 1, m), rng.uniform(-1, 1, m)
    alpha, beta = 1.7, -0.4
    o = oracle.dcsr_optimize(m, m, len(v), 0, rp, ci, v)
    D = _dense(m, m, o["ptr"], o["ind"], o["val"])
    tri = np.tril(D, -1) if fill == P.FILL_LOWER else np.triu(D, 1)
    dg = np.diag(np.diag(D)) if diag == P.DIAG_NON_UNIT else (np.eye(m) if diag == P.DIAG_UNIT else 0.0)
    absx = np.abs(x)
    for mtype, ops in ((P.TYPE_SYMMETRIC, (P.OP_NONE, P.OP_TRANSPOSE)), (P.TYPE_TRIANGULAR, (P.OP_NONE, P.OP_TRANSPOSE))):
        d = P.Descr(mtype=mtype, fill=fill, diag=diag)
        for op in ops:
            st, y = run_dmv(A, d, x, y0, alpha, beta, op=op)
            assert st == 0, (mtype, op, P.STATUS[st])
            if mtype == P.TYPE_SYMMETRIC:
                M = tri + tri.T + dg
                so, yo = oracle.dcsrmv_special("symm", o["base"], alpha, m, m, diag, fill, o["val"], o["ind"],
                                               o["ptr"], o["idiag"], o["iurow"], x, beta, y0)
            else:
                M = tri + dg if op == P.OP_NONE else (tri + dg).T
                so, yo = oracle.dcsrmv_special("tri" if op == P.OP_NONE else "tri_t", o["base"], alpha, m, m, diag,
                                               fill, o["val"], o["ind"], o["ptr"], o["idiag"], o["iurow"], x, beta, y0)
            assert so == 0
            lens = (M != 0).sum(axis=1)
            bound = (lens + 6) * EPS64 * (abs(alpha) * (np.abs(M) @ absx) + np.abs(beta * y0)) + 1e-300
            assert np.all(np.abs(y - yo) <= bound), (mtype, op, np.max(np.abs(y - yo) / bound))
            assert np.all(np.abs(y - (alpha * (M @ x) + beta * y0)) <= 4 * bound)


def test_symmetric_csrmm():
    """csrmm.hpp:667-718 (serial *_sym_ref kernels): C = alpha*(L+D+L^T)*B + beta*C."""
    m, n = 800, 24
    rp, ci, v = random_csr(131, m, m, lambda r, i: r.integers(0, 12))
    A = P.Matrix(0, m, m, rp, ci, v)
    d = P.Descr(mtype=P.TYPE_SYMMETRIC, fill=P.FILL_UPPER)
    rng = np.random.default_rng(3)
    B, C0 = rng.uniform(-1, 1, m * n), rng.uniform(-1, 1, m * n)
    C = C0.copy()
    assert P.dcsrmm(P.OP_NONE, 2.0, A, d, P.ORDER_ROW, B, n, n, 0.5, C, n) == 0
    o = oracle.dcsr_optimize(m, m, len(v), 0, rp, ci, v)
    D = _dense(m, m, o["ptr"], o["ind"], o["val"])
    M = np.triu(D, 1) + np.triu(D, 1).T + np.diag(np.diag(D))
    ref = 2.0 * (M @ B.reshape(m, n)) + 0.5 * C0.reshape(m, n)
    scale = 2.0 * (np.abs(M) @ np.abs(B.reshape(m, n))) + np.abs(0.5 * C0.reshape(m, n))
    assert np.all(np.abs(C.reshape(m, n) - ref) <= 40 * EPS64 * scale + 1e-300)


# --------------------------------------------------------------------------------------------------
# sp2m / spmm / csr2m (spgemm_kernels.hip): structure AND values bit-identical to the reference's
# two-stage Gustavson (first-touch column order, accumulation in visit order)
# --------------------------------------------------------------------------------------------------
def _export(h, double=True):
    base, m, n, nnz = ctypes.c_int(), ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
    rp, ci, v = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
    fn = L.aoclsparse_export_dcsr if double else L.aoclsparse_export_scsr
    st = fn(h, ctypes.byref(base), ctypes.byref(m), ctypes.byref(n), ctypes.byref(nnz), ctypes.byref(rp),
            ctypes.byref(ci), ctypes.byref(v))
    assert st == 0
    ct = ctypes.c_double if double else ctypes.c_float
    k = max(nnz.value, 1)
    row = np.ctypeslib.as_array(ctypes.cast(rp, ctypes.POINTER(ctypes.c_int32)), (m.value + 1,)).copy()
    col = np.ctypeslib.as_array(ctypes.cast(ci, ctypes.POINTER(ctypes.c_int32)), (k,))[: nnz.value].copy()
    val = np.ctypeslib.as_array(ctypes.cast(v, ctypes.POINTER(ct)), (k,))[: nnz.value].copy()
    return base.value, m.value, n.value, nnz.value, row, col, val


@pytest.mark.parametrize("base_a,base_b", [(0, 0), (1, 0), (0, 1), (1, 1)])
def test_sp2m_bit_exact_vs_oracle(base_a, base_b):
    m, k, n = 3000, 2500, 2800
    pa, ia, va = random_csr(201, m, k, lambda r, i: r.integers(0, 9), base=base_a)
    pb, ib, vb = random_csr(202, k, n, lambda r, i: r.integers(0, 9), base=base_b)
    A, B = P.Matrix(base_a, m, k, pa, ia, va), P.Matrix(base_b, k, n, pb, ib, vb)
    dA, dB = P.Descr(base=base_a), P.Descr(base=base_b)
    so, pc, ic, vc = oracle.dcsr2m(m, n, base_a, pa, ia, va, base_b, pb, ib, vb)
    assert so == 0
    # full computation
    C = ctypes.c_void_p()
    assert L.aoclsparse_sp2m(P.OP_NONE, dA.h, A.h, P.OP_NONE, dB.h, B.h, P.STAGE_FULL, ctypes.byref(C)) == 0
    b, cm, cn, cz, row, col, val = _export(C)
    assert (b, cm, cn, cz) == (0, m, n, len(ic))
    assert np.array_equal(row, pc) and np.array_equal(col, ic) and np.array_equal(val, vc)
    assert L.aoclsparse_destroy(ctypes.byref(C)) == 0
    # two-stage protocol: count allocates + fills row_ptr, finalize fills the same handle
    C = ctypes.c_void_p()
    assert L.aoclsparse_dcsr2m(P.OP_NONE, dA.h, A.h, P.OP_NONE, dB.h, B.h, P.STAGE_NNZ_COUNT, ctypes.byref(C)) == 0
    h0 = C.value
    b, cm, cn, cz, row, _, _ = _export(C)
    assert cz == len(ic) and np.array_equal(row, pc)
    assert L.aoclsparse_dcsr2m(P.OP_NONE, dA.h, A.h, P.OP_NONE, dB.h, B.h, P.STAGE_FINALIZE, ctypes.byref(C)) == 0
    assert C.value == h0
    _, _, _, _, row, col, val = _export(C)
    assert np.array_equal(col, ic) and np.array_equal(val, vc)
    assert L.aoclsparse_destroy(ctypes.byref(C)) == 0


def test_sp2m_transposes_and_long_rows():
    """op(A), op(B) in {N, T}; one operand row long enough (upper bound > 1024) for the global-slab path."""
    m, k, n = 400, 350, 500
    pa, ia, va = random_csr(211, m, k, lambda r, i: 300 if i == 7 else r.integers(0, 7))
    pb, ib, vb = random_csr(212, k, n, lambda r, i: r.integers(0, 12))
    A, B = P.Matrix(0, m, k, pa, ia, va), P.Matrix(0, k, n, pb, ib, vb)
    d = P.Descr()
    DA, DB = _dense(m, k, pa, ia, va), _dense(k, n, pb, ib, vb)
    so, pc, ic, vc = oracle.dcsr2m(m, n, 0, pa, ia, va, 0, pb, ib, vb)
    C = ctypes.c_void_p()
    assert L.aoclsparse_spmm(P.OP_NONE, A.h, B.h, ctypes.byref(C)) == 0
    _, _, _, cz, row, col, val = _export(C)
    assert np.array_equal(row, pc) and np.array_equal(col, ic) and np.array_equal(val, vc)
    assert np.max(np.diff(pc)) > 0 and (pa[8] - pa[7]) == 300
    L.aoclsparse_destroy(ctypes.byref(C))
    # A^T * B2, A * B3^T, A^T * B4^T against dense products and the oracle on explicit transposes
    st, tp, ti, tv = oracle.dcsr2csc(m, k, len(va), 0, 0, pa, ia, va)  # A^T as CSR (k x m)
    pb2, ib2, vb2 = random_csr(213, m, n, lambda r, i: r.integers(0, 10))
    B2 = P.Matrix(0, m, n, pb2, ib2, vb2)
    assert L.aoclsparse_sp2m(P.OP_TRANSPOSE, d.h, A.h, P.OP_NONE, d.h, B2.h, P.STAGE_FULL, ctypes.byref(C)) == 0
    _, cm, cn, cz, row, col, val = _export(C)
    so, pc, ic, vc = oracle.dcsr2m(k, n, 0, tp, ti, tv, 0, pb2, ib2, vb2)
    assert (cm, cn) == (k, n) and np.array_equal(row, pc) and np.array_equal(col, ic) and np.array_equal(val, vc)
    L.aoclsparse_destroy(ctypes.byref(C))
    pb3, ib3, vb3 = random_csr(214, n, k, lambda r, i: r.integers(0, 10))
    B3 = P.Matrix(0, n, k, pb3, ib3, vb3)
    assert L.aoclsparse_sp2m(P.OP_NONE, d.h, A.h, P.OP_TRANSPOSE, d.h, B3.h, P.STAGE_FULL, ctypes.byref(C)) == 0
    _, cm, cn, cz, row, col, val = _export(C)
    assert (cm, cn) == (m, n)
    assert np.allclose(_dense(cm, cn, row, col, val), DA @ _dense(n, k, pb3, ib3, vb3).T, atol=1e-12)
    L.aoclsparse_destroy(ctypes.byref(C))
    pb4, ib4, vb4 = random_csr(215, n, m, lambda r, i: r.integers(0, 10))
    B4 = P.Matrix(0, n, m, pb4, ib4, vb4)
    assert L.aoclsparse_sp2m(P.OP_TRANSPOSE, d.h, A.h, P.OP_TRANSPOSE, d.h, B4.h, P.STAGE_FULL, ctypes.byref(C)) == 0
    _, cm, cn, cz, row, col, val = _export(C)
    assert (cm, cn) == (k, n)
    assert np.allclose(_dense(cm, cn, row, col, val), DA.T @ _dense(n, m, pb4, ib4, vb4).T, atol=1e-12)
    assert all(np.all(np.diff(col[row[i]:row[i + 1]]) > 0) for i in range(cm))  # transposed back: sorted rows
    L.aoclsparse_destroy(ctypes.byref(C))


def test_sp2m_laplacian_squared_and_float():
    """A*A on L100 has 128,004 non-zeros."""
    m, rp, ci, v = laplace5(100)
    A = P.Matrix(0, m, m, rp, ci, v)
    d = P.Descr()
    C = ctypes.c_void_p()
    assert L.aoclsparse_sp2m(P.OP_NONE, d.h, A.h, P.OP_NONE, d.h, A.h, P.STAGE_FULL, ctypes.byref(C)) == 0
    _, cm, cn, cz, row, col, val = _export(C)
    assert cz == 128004
    so, pc, ic, vc = oracle.dcsr2m(m, m, 0, rp, ci, v, 0, rp, ci, v)
    assert np.array_equal(row, pc) and np.array_equal(col, ic) and np.array_equal(val, vc)
    L.aoclsparse_destroy(ctypes.byref(C))
    Af = P.Matrix(0, m, m, rp, ci, v.astype(np.float32))
    assert L.aoclsparse_scsr2m(P.OP_NONE, d.h, Af.h, P.OP_NONE, d.h, Af.h, P.STAGE_FULL, ctypes.byref(C)) == 0
    _, _, _, cz, row, col, val = _export(C, double=False)
    assert cz == 128004 and np.array_equal(col, ic) and np.array_equal(val, vc.astype(np.float32))
    L.aoclsparse_destroy(ctypes.byref(C))


# --------------------------------------------------------------------------------------------------
# extreme values and concurrency (mv_tests.cpp:1858-2290, context_tests.cpp:237-357)
# --------------------------------------------------------------------------------------------------
def test_nan_inf_propagation_matches_reference_arithmetic():
    """NaN / Inf in A or x propagate exactly as the reference's FMA chains propagate them."""
    m = n = 2000
    for rl, seed in ((lambda r, i: r.integers(0, 9), 301), (lambda r, i: r.integers(0, 60), 302)):
        rp, ci, v = random_csr(seed, m, n, rl)
        x = np.random.default_rng(1).uniform(-1, 1, n)
        v = v.copy()
        v[5], v[17], v[100] = np.inf, -np.inf, np.nan
        x[3], x[40], x[77] = np.nan, np.inf, 1e308
        v[200] = 1e308
        y0 = np.random.default_rng(2).uniform(-1, 1, m)
        A = P.Matrix(0, m, n, rp, ci, v)
        d = P.Descr()
        for alpha, beta in ((1.0, 0.0), (2.0, 0.5)):
            st, y = run_dmv(A, d, x, y0, alpha, beta)
            so, yr = oracle.dcsrmv(-1, 0, alpha, m, len(v), v, ci, rp, x, beta, y0)
            assert st == 0 and np.array_equal(np.isnan(y), np.isnan(yr))
            ok = ~np.isnan(yr)
            assert np.array_equal(y[ok], yr[ok]) and np.isnan(y).sum() > 0 and np.isinf(y).sum() > 0


def test_concurrent_executors_on_one_handle():
    """Executors may be called from several host threads on the same handle (threading)."""
    import threading

    m, rp, ci, v = laplace5(300)
    A = P.Matrix(0, m, m, rp, ci, v)
    d = P.Descr()
    rng = np.random.default_rng(5)
    xs = [rng.uniform(-1, 1, m) for _ in range(4)]
    outs = [None] * 4

    def work(k):
        for _ in range(10):  # host-pointer path (staging buffers are shared: must serialise correctly)
            y = np.zeros(m)
            assert P.dmv(P.OP_NONE, 1.0, A, d, xs[k], 0.0, y) == 0
            outs[k] = y

    th = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    [t.start() for t in th]
    [t.join() for t in th]
    for k in range(4):
        so, yr = oracle.dcsrmv(-1, 0, 1.0, m, len(v), v, ci, rp, xs[k], 0.0, np.zeros(m))
        assert np.array_equal(outs[k], yr)


def test_csrmm_beta0_nonfinite_c_policy():
    """beta == 0.  Default (round 3): C is read and multiplied by zero as in every reference kernel (csrmm.hpp:83,129;
    csrmm_kt.cpp:176-191,246), so the result equals the oracle bit for bit for finite C (signed zeros of empty rows
    included) AND a NaN / Inf already in C propagates exactly where the oracle's does.  Opt-in
    (aoclsparse_mi355_set_csrmm_beta0_overwrite(1)): C is not read; identical bits for finite C, non-finite C overwritten."""
    from util import beta0_overwrite

    m, k, n = 300, 300, 8
    rp, ci, v = random_csr(141, m, k, lambda r, i: 0 if i % 5 == 0 else r.integers(1, 6))
    rng = np.random.default_rng(4)
    B = rng.uniform(-1, 1, k * n)
    C0 = rng.uniform(-1, 1, m * n)  # finite, both signs: decides the sign of the zeros of empty rows
    A = P.Matrix(0, m, k, rp, ci, v)
    d = P.Descr()

    def reference(alpha, oname, ldb, ldc, Cin):
        if oname == "col":
            return oracle.dcsrmm("col", alpha, 0, v, ci, rp, m, B, n, ldb, 0.0, Cin, ldc)[1]
        # same per-element arithmetic on the transposed layout
        Cr = oracle.dcsrmm("col", alpha, 0, v, ci, rp, m, np.ascontiguousarray(B.reshape(k, n).T).ravel(), n, k, 0.0,
                           np.ascontiguousarray(Cin.reshape(m, n).T).ravel(), m)[1]
        return np.ascontiguousarray(Cr.reshape(n, m).T).ravel()

    Cn = C0.copy()
    Cn[::7] = np.nan
    Cn[3::11] = np.inf
    for overwrite in (False, True):
        if overwrite:
            assert L.aoclsparse_mi355_set_csrmm_beta0_overwrite(1) == 0
        try:
            for alpha in (1.5, -1.5):
                for order, ldb, ldc, oname in ((P.ORDER_COLUMN, k, m, "col"), (P.ORDER_ROW, n, n, "row")):
                    for on_device in (False, True):
                        C = dev(C0) if on_device else C0.copy()
                        assert P.dcsrmm(P.OP_NONE, alpha, A, d, order, dev(B) if on_device else B, n, ldb, 0.0, C, ldc) == 0
                        got = C.cpu().numpy() if on_device else C
                        Cr = reference(alpha, oname, ldb, ldc, C0)
                        assert np.array_equal(got, Cr) and np.array_equal(np.signbit(got), np.signbit(Cr)), (overwrite, oname)
                        # non-finite C
                        C = dev(Cn) if on_device else Cn.copy()
                        assert P.dcsrmm(P.OP_NONE, alpha, A, d, order, dev(B) if on_device else B, n, ldb, 0.0, C, ldc) == 0
                        got = C.cpu().numpy() if on_device else C
                        if not overwrite:
                            Cr = reference(alpha, oname, ldb, ldc, Cn)   # 0 * NaN = 0 * Inf = NaN, as in the reference
                            assert np.array_equal(np.isnan(got), np.isnan(Cr)) and np.isnan(got).sum() > 0
                            assert np.array_equal(got[~np.isnan(got)], Cr[~np.isnan(Cr)])
                        else:
                            Cz = reference(alpha, oname, ldb, ldc, np.zeros(m * n))
                            nz = Cz != 0
                            assert np.array_equal(got[nz], Cz[nz])  # overwritten where the product is non-zero
        finally:
            assert L.aoclsparse_mi355_set_csrmm_beta0_overwrite(0) == 0


# --------------------------------------------------------------------------------------------------
# "next" rows of SURVEY 8f: trsm, dotmv, value mutation
# --------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("order", [P.ORDER_ROW, P.ORDER_COLUMN])
@pytest.mark.parametrize("kid", [None, 0, 3])
def test_trsm_equals_trsv_per_column(order, kid):
    """level3/aoclsparse_trsm.hpp:150-158: trsm IS a loop of trsv over the columns -> each column must be
    bit-identical to the serial reference solve of that column."""
    m, n = 6000, 7
    rp, ci, v = triangular_system(161, m, 5, band=200)
    A = P.Matrix(0, m, m, rp, ci, v)
    rng = np.random.default_rng(6)
    for fill, trans, unit in (("lower", "n", False), ("upper", "t", True)):
        d = P.Descr(mtype=P.TYPE_TRIANGULAR, fill=P.FILL_LOWER if fill == "lower" else P.FILL_UPPER,
                    diag=P.DIAG_UNIT if unit else P.DIAG_NON_UNIT)
        op = P.OP_NONE if trans == "n" else P.OP_TRANSPOSE
        if order == P.ORDER_ROW:
            ldb, ldx = n + 2, n + 1
            Bm, Xm = rng.uniform(-1, 1, (m, ldb)), np.full((m, ldx), 9.0)
            cols_b = [Bm[:, j].copy() for j in range(n)]
        else:
            ldb, ldx = m + 3, m
            Bm, Xm = rng.uniform(-1, 1, (n, ldb)), np.full((n, ldx), 9.0)
            cols_b = [Bm[j, :m].copy() for j in range(n)]
        fn = L.aoclsparse_dtrsm if kid is None else L.aoclsparse_dtrsm_kid
        args = [op, 0.7, A.h, d.h, order, P._ptr(Bm), n, ldb, P._ptr(Xm), ldx] + ([] if kid is None else [kid])
        assert fn(*args) == 0
        Xd = dev(np.full_like(Xm, 9.0))
        args[5], args[8] = P._ptr(dev(Bm)), P._ptr(Xd)
        assert fn(*args) == 0
        torch.cuda.synchronize()
        assert np.array_equal(Xd.cpu().numpy(), Xm)
        for j in range(n):
            xr = oracle_trsv(0, m, rp, ci, v, fill, trans, unit, 0.7, cols_b[j], kid=kid)
            got = Xm[:, j] if order == P.ORDER_ROW else Xm[j, :m]
            assert np.array_equal(got, xr), (order, kid, fill, j)
        if order == P.ORDER_ROW:
            assert np.all(Xm[:, n:] == 9.0)  # padding untouched


def test_dotmv():
    m, n = 3000, 2600
    rp, ci, v = random_csr(171, m, n, lambda r, i: r.integers(0, 12))
    A = P.Matrix(0, m, n, rp, ci, v)
    d = P.Descr()
    rng = np.random.default_rng(8)
    x, y0 = rng.uniform(-1, 1, n), rng.uniform(-1, 1, m)
    y, dot = y0.copy(), np.zeros(1)
    assert L.aoclsparse_ddotmv(P.OP_NONE, 1.3, A.h, d.h, P._ptr(x), -0.2, P._ptr(y), P._ptr(dot)) == 0
    so, yr = oracle.dcsrmv(-1, 0, 1.3, m, len(v), v, ci, rp, x, -0.2, y0)
    assert np.array_equal(y, yr)
    k = min(m, n)
    ref = float(np.dot(x[:k].astype(np.longdouble), yr[:k].astype(np.longdouble)))
    assert abs(dot[0] - ref) <= 2 * k * EPS64 * float(np.dot(np.abs(x[:k]), np.abs(yr[:k])))
    xd, yd, dd = dev(x), dev(y0), torch.zeros(1, dtype=torch.float64, device="cuda")
    assert L.aoclsparse_ddotmv(P.OP_NONE, 1.3, A.h, d.h, P._ptr(xd), -0.2, P._ptr(yd), P._ptr(dd)) == 0
    torch.cuda.synchronize()
    assert np.array_equal(yd.cpu().numpy(), yr) and dd.item() == dot[0]  # deterministic tree


def test_set_value_and_update_values_refresh_the_device_copy():
    m, rp, ci, v = laplace5(40)
    v = v.copy()
    A = P.Matrix(0, m, m, rp, ci, v)
    d = P.Descr()
    x = np.random.default_rng(3).uniform(-1, 1, m)
    st, y1 = run_dmv(A, d, x, np.zeros(m), 1.0, 0.0)
    assert L.aoclsparse_dset_value(A.h, 7, 7, 123.5) == 0
    st, y2 = run_dmv(A, d, x, np.zeros(m), 1.0, 0.0)
    so, yr = oracle.dcsrmv(-1, 0, 1.0, m, len(v), A.val, ci, rp, x, 0.0, np.zeros(m))
    assert np.array_equal(y2, yr) and y2[7] != y1[7]
    nv = np.random.default_rng(4).uniform(-1, 1, len(v))
    assert L.aoclsparse_dupdate_values(A.h, len(v), P._ptr(nv)) == 0
    st, y3 = run_dmv(A, d, x, np.zeros(m), 1.0, 0.0)
    so, yr = oracle.dcsrmv(-1, 0, 1.0, m, len(v), nv, ci, rp, x, 0.0, np.zeros(m))
    assert np.array_equal(y3, yr)
    # the triangular solve plan is rebuilt from the new values too
    dt = P.Descr(mtype=P.TYPE_TRIANGULAR, fill=P.FILL_LOWER, diag=P.DIAG_UNIT)
    b, xs = np.ones(m), np.zeros(m)
    assert P.dtrsv(P.OP_NONE, 1.0, A, dt, b, xs) == 0
    assert L.aoclsparse_dset_value(A.h, 50, 49, 0.25) == 0
    xs2 = np.zeros(m)
    assert P.dtrsv(P.OP_NONE, 1.0, A, dt, b, xs2) == 0
    assert np.array_equal(xs2, oracle_trsv(0, m, rp, ci, A.val, "lower", "n", True, 1.0, b)) and not np.array_equal(xs, xs2)


def test_symgs_reference_kats(kats):
    """symgs_tests.cpp:380-455: every system, fill mode, operation, index base; ?symgs and ?symgs_mv; host and
    device vectors.  Tolerance = the reference's expected_precision(10)."""
    tol = kats["trsv_abs_tol"]
    mt = {"general": P.TYPE_GENERAL, "symmetric": P.TYPE_SYMMETRIC}
    for c in kats["symgs"]:
        m = c["m"]
        for base in (0, 1):
            rp = np.array(c["row_ptr"], np.int32) + base
            ci = np.array(c["col_ind"], np.int32) + base
            v = np.array(c["val"], np.float64)
            b = np.array(c["b"], np.float64)
            A = P.Matrix(base, m, m, rp, ci, v)
            for fill in (P.FILL_LOWER, P.FILL_UPPER):
                for ti, op in enumerate((P.OP_NONE, P.OP_TRANSPOSE)):
                    d = P.Descr(base=base, mtype=mt[c["mtype"]], fill=fill)
                    xg = c["x_gold"] if c["mtype"] == "symmetric" else c["x_gold"]["nt"[ti]]
                    yg = c["y_gold"] if c["mtype"] == "symmetric" else c["y_gold"]["nt"[ti]]
                    x, y = np.array(c["x0"], np.float64), np.zeros(m)
                    for _ in range(c["iters"]):
                        assert L.aoclsparse_dsymgs_mv(op, A.h, d.h, c["alpha"], P._ptr(b), P._ptr(x), P._ptr(y)) == 0
                    assert np.all(np.abs(x - np.array(xg)) <= tol), (c["name"], base, fill, ti)
                    assert np.all(np.abs(y - np.array(yg)) <= tol * max(1.0, np.abs(yg).max())), (c["name"], base, fill, ti)
                    xd, bd = dev(np.array(c["x0"], np.float64)), dev(b)
                    for _ in range(c["iters"]):
                        assert L.aoclsparse_dsymgs_kid(op, A.h, d.h, c["alpha"], P._ptr(bd), P._ptr(xd), 0) == 0
                    torch.cuda.synchronize()
                    assert np.array_equal(xd.cpu().numpy(), x)


@pytest.mark.parametrize("mtype", ["symmetric", "general"])
def test_symgs_large_against_oracle(mtype):
    """One sweep on a 2-D Laplacian-like SPD matrix with random off-diagonals; the TRSV stages are bit-exact,
    the triangular products carry the SpMV bound, so compare within a small multiple of eps * scale."""
    g = 70
    m, rp, ci, v = laplace5(g)
    rng = np.random.default_rng(17)
    v = v.copy()
    if mtype == "general":
        v[v < 0] = rng.uniform(-1.0, -0.2, np.count_nonzero(v < 0))
    A = P.Matrix(0, m, m, rp, ci, v)
    b, x0 = rng.uniform(-1, 1, m), rng.uniform(-1, 1, m)
    o = oracle.dcsr_optimize(m, m, len(v), 0, rp, ci, v)
    for fill, ti in ((0, 0), (1, 0), (0, 1)):
        d = P.Descr(mtype=P.TYPE_SYMMETRIC if mtype == "symmetric" else P.TYPE_GENERAL,
                    fill=P.FILL_LOWER if fill == 0 else P.FILL_UPPER)
        x = x0.copy()
        assert L.aoclsparse_dsymgs((P.OP_NONE, P.OP_TRANSPOSE)[ti], A.h, d.h, 0.9, P._ptr(b), P._ptr(x)) == 0
        st, xr = oracle.dsymgs(1 if mtype == "symmetric" else 0, fill, ti, 0, 0.9, m, o["val"], o["ind"], o["ptr"],
                               o["idiag"], o["iurow"], b, x0)
        assert st == 0
        assert np.max(np.abs(x - xr)) <= 64 * EPS64 * max(1.0, np.abs(xr).max()), (fill, ti, np.max(np.abs(x - xr)))


def test_symgs_and_ilu_argument_checks_on_gpu_box():
    m, rp, ci, v = laplace5(6)
    A = P.Matrix(0, m, m, rp, ci, v)
    b, x = np.ones(m), np.ones(m)
    assert L.aoclsparse_dsymgs(P.OP_NONE, A.h, P.Descr(mtype=P.TYPE_SYMMETRIC, diag=P.DIAG_UNIT).h, 1.0, P._ptr(b), P._ptr(x)) == 1
    assert L.aoclsparse_dsymgs(113, A.h, P.Descr().h, 1.0, P._ptr(b), P._ptr(x)) == 1  # general + conj. transpose
    assert L.aoclsparse_dsymgs(P.OP_NONE, A.h, P.Descr(base=1).h, 1.0, P._ptr(b), P._ptr(x)) == 5
    assert L.aoclsparse_ssymgs(P.OP_NONE, A.h, P.Descr().h, 1.0, P._ptr(b), P._ptr(x)) == 9
    assert L.aoclsparse_dsymgs(P.OP_NONE, A.h, P.Descr().h, 1.0, None, P._ptr(x)) == 2
    assert L.aoclsparse_dsymgs_mv(P.OP_NONE, A.h, P.Descr().h, 1.0, P._ptr(b), P._ptr(x), None) == 2


@pytest.mark.parametrize("base", [0, 1])
def test_ilu_smoother_matches_reference_factorisation_and_solve(base):
    """ilu_tests.cpp drives aoclsparse_dilu_smoother on a handle with an LU-smoother hint: the factors returned
    through precond_csr_val and the smoothed x must equal the serial IKJ factorisation + the two serial solves."""
    g = 48
    m, rp, ci, v = laplace5(g)
    rng = np.random.default_rng(23)
    v = v * rng.uniform(0.8, 1.2, len(v))
    rp, ci = rp + base, ci + base
    A = P.Matrix(base, m, m, rp, ci, v)
    d = P.Descr(base=base)
    assert L.aoclsparse_set_lu_smoother_hint(A.h, P.OP_NONE, d.h, 10) == 0 and L.aoclsparse_optimize(A.h) == 0
    st, lu, diag = oracle.dilu0(m, base, rp, ci, v)
    assert st == 0
    pv = ctypes.c_void_p()
    for it in range(2):  # second call reuses the factors
        b, x = rng.uniform(-1, 1, m), np.zeros(m)
        assert L.aoclsparse_dilu_smoother(P.OP_NONE, A.h, d.h, ctypes.byref(pv), None, P._ptr(x), P._ptr(b)) == 0
        fac = np.ctypeslib.as_array(ctypes.cast(pv, ctypes.POINTER(ctypes.c_double)), (len(v),))
        assert np.array_equal(fac, lu)
        st, xr = oracle.dilu_solve(m, base, diag, lu, rp, ci, b)
        assert st == 0 and np.array_equal(x, xr)
        xd = torch.zeros(m, dtype=torch.float64, device="cuda")
        assert L.aoclsparse_dilu_smoother(P.OP_NONE, A.h, d.h, ctypes.byref(pv), None, P._ptr(xd), P._ptr(dev(b))) == 0
        torch.cuda.synchronize()
        assert np.array_equal(xd.cpu().numpy(), xr)
    assert np.array_equal(A.val, v)  # the user's values are never touched
    assert L.aoclsparse_dilu_smoother(P.OP_TRANSPOSE, A.h, d.h, ctypes.byref(pv), None, P._ptr(x), P._ptr(b)) == 1
    assert L.aoclsparse_dilu_smoother(P.OP_NONE, A.h, P.Descr(base=base, mtype=P.TYPE_SYMMETRIC).h, ctypes.byref(pv), None,
                                      P._ptr(x), P._ptr(b)) == 1


@pytest.mark.parametrize("base", [0, 1])
def test_ilu_smoother_sync_free_factorisation_long_rows(base):
    """matrices with >= 16 entries per row are factorised in ONE sync-free launch (rows wait for the rows they need through
    their diagonal-position word): factors bit-identical to the serial IKJ loop, both precisions; a zero pivot fails the
    call (and every row that depends on the failed one) instead of hanging"""
    from test_gpu_trsv_blocks import node_mesh
    nodes = 1800
    m, rp, ci, v = node_mesh(81, nodes, 36, np.full(nodes, 5), keep=0.95)
    assert len(v) >= 16 * m
    rp, ci = rp + base, ci + base
    d = P.Descr(base=base)
    rng = np.random.default_rng(82)
    st, lu, diag = oracle.dilu0(m, base, rp, ci, v)
    assert st == 0
    A = P.Matrix(base, m, m, rp, ci, v)
    pv = ctypes.c_void_p()
    b, x = rng.uniform(-1, 1, m), np.zeros(m)
    assert L.aoclsparse_dilu_smoother(P.OP_NONE, A.h, d.h, ctypes.byref(pv), None, P._ptr(x), P._ptr(b)) == 0
    fac = np.ctypeslib.as_array(ctypes.cast(pv, ctypes.POINTER(ctypes.c_double)), (len(v),))
    assert np.array_equal(fac, lu)
    st, xr = oracle.dilu_solve(m, base, diag, lu, rp, ci, b)
    assert st == 0 and np.array_equal(x, xr)
    # float
    vf = v.astype(np.float32)
    Af = P.Matrix(base, m, m, rp, ci, vf)
    pf = ctypes.c_void_p()
    xf = np.zeros(m, np.float32)
    assert L.aoclsparse_silu_smoother(P.OP_NONE, Af.h, d.h, ctypes.byref(pf), None, P._ptr(xf), P._ptr(b.astype(np.float32))) == 0
    facf = np.ctypeslib.as_array(ctypes.cast(pf, ctypes.POINTER(ctypes.c_float)), (len(v),)).copy()
    lu32 = vf.copy()  # serial IKJ in fp32 (the reference's loop, ilu0.hpp:34-111), first 400 rows
    pos = [dict((int(c), int(p)) for p, c in zip(range(rp[i] - base, rp[i + 1] - base), ci[rp[i] - base:rp[i + 1] - base] - base)) for i in range(400)]
    dg = {}
    for i in range(400):
        for p in range(rp[i] - base, rp[i + 1] - base):
            k = int(ci[p] - base)
            if k >= i:
                break
            lik = np.float32(lu32[p] / lu32[dg[k]])
            lu32[p] = lik
            for q in range(dg[k] + 1, rp[k + 1] - base):
                w = pos[i].get(int(ci[q] - base))
                if w is not None:
                    lu32[w] = np.float32(np.float64(lu32[w]) - np.float64(lik) * np.float64(lu32[q]))  # = fmaf: exact product, one rounding
        dg[i] = pos[i][i]
    n400 = rp[400] - base
    assert np.array_equal(facf[:n400], lu32[:n400])
    # a zero pivot in row 0 (nothing updates it before it is used): numerical error, no hang
    vz = v.copy()
    for p in range(rp[0] - base, rp[1] - base):
        if ci[p] - base == 0:
            vz[p] = 0.0
    assert oracle.dilu0(m, base, rp, ci, vz)[0] != 0
    Az = P.Matrix(base, m, m, rp, ci, vz)
    assert L.aoclsparse_dilu_smoother(P.OP_NONE, Az.h, d.h, ctypes.byref(pv), None, P._ptr(x), P._ptr(b)) == 11  # numerical_error


# --------------------------------------------------------------------------------------------------
# ELL family
# --------------------------------------------------------------------------------------------------
def _ell_rows(seed, m, n, maxlen, long_every=0):
    rng = np.random.default_rng(seed)
    lens = rng.integers(0, maxlen, m)
    lens[min(5, m - 1)] = 0
    if long_every:
        lens[::long_every] = rng.integers(3 * maxlen, 9 * maxlen, len(lens[::long_every]))
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    ci = np.concatenate([np.sort(rng.choice(n, k, replace=False)) for k in lens]).astype(np.int32)
    return rp, ci, rng.uniform(-1, 1, len(ci))


def test_ellmv_reference_kat(kats):
    c = kats["ell"][0]
    d = P.Descr(base=c["base"])
    col, val, x = np.array(c["ell_col_ind"], np.int32), np.array(c["ell_val"]), np.array(c["x"])
    y, a, b = np.full(3, np.nan), np.array([c["alpha"]]), np.array([c["beta"]])
    assert L.aoclsparse_dellmv(P.OP_NONE, P._ptr(a), 3, 3, 4, P._ptr(val), P._ptr(col), 2, d.h, P._ptr(x), P._ptr(b), P._ptr(y)) == 0
    assert list(y) == c["y_gold"]
    yf = np.full(3, np.nan, np.float32)
    assert L.aoclsparse_sellmv(P.OP_NONE, P._ptr(a.astype(np.float32)), 3, 3, 4, P._ptr(val.astype(np.float32)), P._ptr(col), 2,
                               d.h, P._ptr(x.astype(np.float32)), P._ptr(b.astype(np.float32)), P._ptr(yf)) == 0
    assert list(yf) == c["y_gold"]


@pytest.mark.parametrize("base", [0, 1])
@pytest.mark.parametrize("alpha,beta", [(1.0, 0.0), (1.7, -0.3)])
def test_ell_family_bit_exact(base, alpha, beta):
    """?ellmv (4-lane order / scalar float), ?elltmv, dellthybmv against the restated reference kernels, host and
    device arrays, both index bases; beta == 0 must not read y."""
    m, n = 5000, 4700
    rp, ci, v = _ell_rows(51, m, n, 23, long_every=97)
    rp, ci = rp + base, ci + base
    d = P.Descr(base=base)
    rng = np.random.default_rng(52)
    x = rng.uniform(-1, 1, n)
    y0 = rng.uniform(-1, 1, m) if beta != 0.0 else np.full(m, np.nan)
    a, b = np.array([alpha]), np.array([beta])
    for layout, fn in (("ell", L.aoclsparse_dellmv), ("ellt", L.aoclsparse_delltmv)):
        w, ec, ev = oracle.csr2ell(layout, m, base, rp, ci, v)
        st, yr = oracle.dellmv(layout, base, alpha, m, ev, ec, w, x, beta, y0)
        y = y0.copy()
        assert fn(P.OP_NONE, P._ptr(a), m, n, len(v), P._ptr(ev), P._ptr(ec), w, d.h, P._ptr(x), P._ptr(b), P._ptr(y)) == 0
        assert np.array_equal(y, yr), layout
        yd, evd, ecd = dev(y0), dev(ev), dev(ec)
        assert fn(P.OP_NONE, P._ptr(a), m, n, len(v), P._ptr(evd), P._ptr(ecd), w, d.h, P._ptr(dev(x)), P._ptr(b), P._ptr(yd)) == 0
        torch.cuda.synchronize()
        assert np.array_equal(yd.cpu().numpy(), yr), layout
    w, em, mp, hc, hv = oracle.csr2ell("hyb", m, base, rp, ci, v)
    assert 0 < len(mp) < m
    st, yr = oracle.dellthybmv(base, alpha, m, hv, hc, w, em, v, rp, ci, mp, x, beta, y0)
    y = y0.copy()
    args = [P.OP_NONE, P._ptr(a), m, n, len(v), P._ptr(hv), P._ptr(hc), w, em, P._ptr(v), P._ptr(rp), P._ptr(ci), None, P._ptr(mp),
            d.h, P._ptr(x), P._ptr(b), P._ptr(y)]
    assert L.aoclsparse_dellthybmv(*args) == 0
    assert np.array_equal(y, yr)
    keep = [dev(t) for t in (hv, hc, v, rp, ci, mp, x, y0)]
    for k, t in zip((5, 6, 9, 10, 11, 13, 15, 17), keep):
        args[k] = P._ptr(t)
    assert L.aoclsparse_dellthybmv(*args) == 0
    torch.cuda.synchronize()
    assert np.array_equal(keep[-1].cpu().numpy(), yr)
    # float: scalar-order ELL and ELLT
    vf, xf = v.astype(np.float32), x.astype(np.float32)
    af, bf = a.astype(np.float32), b.astype(np.float32)
    y0f = y0.astype(np.float32)
    w, ec, ev = oracle.csr2ell("ell", m, base, rp, ci, vf.astype(np.float64))
    st, yr = oracle.sellmv(base, float(af[0]), m, ev.astype(np.float32), ec, w, xf, float(bf[0]), y0f)
    yf = y0f.copy()
    assert L.aoclsparse_sellmv(P.OP_NONE, P._ptr(af), m, n, len(v), P._ptr(ev.astype(np.float32)), P._ptr(ec), w, d.h, P._ptr(xf),
                               P._ptr(bf), P._ptr(yf)) == 0
    assert np.array_equal(yf, yr)


def _blkmv(base, alpha, beta, m, n, nnz, mk, bv, bc, brp, x, y, rows, on_device):
    a, b, d = np.array([alpha]), np.array([beta]), P.Descr(base=base)
    if on_device:
        mk, bv, bc, brp, x, yd = dev(mk), dev(bv), dev(bc), dev(brp), dev(x), dev(y)
        st = L.aoclsparse_dblkcsrmv(P.OP_NONE, P._ptr(a), m, n, nnz, P._ptr(mk), P._ptr(bv), P._ptr(bc), P._ptr(brp), d.h,
                                    P._ptr(x), P._ptr(b), P._ptr(yd), rows)
        torch.cuda.synchronize()
        return st, yd.cpu().numpy()
    y = y.copy()
    st = L.aoclsparse_dblkcsrmv(P.OP_NONE, P._ptr(a), m, n, nnz, P._ptr(mk), P._ptr(bv), P._ptr(bc), P._ptr(brp), d.h, P._ptr(x),
                                P._ptr(b), P._ptr(y), rows)
    return st, y


def test_blkcsrmv_reference_kats(kats):
    """blkcsrmv_tests.cpp:444-470 (arrays fed directly) and :518-537 / :656-676 (through csr2blkcsr, 1/2/4 x 8)."""
    c = kats["blkcsr"]["direct"]
    st, y = _blkmv(c["base"], c["alpha"], c["beta"], c["m"], c["n"], c["nnz"], np.array(c["masks"], np.uint8), np.array(c["val"]),
                   np.array(c["blk_col_ind"], np.int32), np.array(c["blk_row_ptr"], np.int32), np.array(c["x"]),
                   np.full(c["m"], np.nan), c["rows_blk"], False)
    assert st == 0 and list(y) == c["y_gold"]
    for c in kats["blkcsr"]["csr"]:
        for rows in (1, 2, 4):
            so, brp, bc, bv, mk = oracle.csr2blkcsr(c["m"], c["n"], c["base"], c["row_ptr"], c["col_ind"], c["val"], rows)
            for on_device in (False, True):
                st, y = _blkmv(c["base"], c["alpha"], c["beta"], c["m"], c["n"], c["nnz"], mk, bv, bc, brp, np.array(c["x"]),
                               np.full(c["m"], np.nan), rows, on_device)
                assert st == 0 and list(y) == c["y_gold"]


@pytest.mark.parametrize("base", [0, 1])
@pytest.mark.parametrize("rows", [1, 2, 4])
def test_blkcsrmv_bit_exact(base, rows):
    """the 8 lanes of a row group are the 8 lanes of the reference's zmm accumulator: bit-identical to the restated
    AVX-512 kernels (blkcsrmv_avx512.cpp:40-369) for any alpha/beta, row counts that are not a multiple of the block
    height, empty rows, windows re-anchored at n-8, and more blocks than one scan chunk (1024)."""
    for seed, m, n, per, alpha, beta in ((1, 3001, 2500, lambda r, i: 0 if i % 13 == 5 else 6 + (i * 7) % 40, 1.0, 0.0),
                                         (2, 777, 21, lambda r, i: 2 + i % 15, -1.25, 0.5),
                                         (3, 50, 8, lambda r, i: 1 + i % 8, 2.0, 1.0)):
        rp, ci, v = banded_rows(seed, m, n, per, base)
        so, brp, bc, bv, mk = oracle.csr2blkcsr(m, n, base, rp, ci, v, rows)
        assert so == 0 and (seed != 1 or len(bc) > 2048)
        rng = np.random.default_rng(seed)
        x = rng.uniform(-1, 1, n)
        y0 = rng.uniform(-1, 1, m) if beta != 0.0 else np.full(m, np.nan)
        so, yr = oracle.dblkcsrmv(base, alpha, m, mk, bv, bc, brp, x, beta, y0, rows)
        for on_device in (False, True):
            st, y = _blkmv(base, alpha, beta, m, n, len(v), mk, bv, bc, brp, x, y0, rows, on_device)
            assert st == 0 and np.array_equal(y, yr)
        # and it is the same product as the CSR one up to the regrouping of each row's terms
        so, yc = oracle.dcsrmv_order("ref", base, alpha, m, v, ci, rp, x, beta, np.nan_to_num(y0))
        scale = abs(alpha) * abs_row_sums(rp, ci, v, x, base) + abs(beta * np.nan_to_num(y0))
        assert np.all(np.abs(yr - yc) <= (np.diff(rp) + 4) * EPS64 * scale + 1e-300)


def test_blkcsrmv_window_reads_x_like_the_expand_load():
    """a lane whose mask bit is clear still multiplies x by 0 (the reference's zero-filled expand-load), so a NaN in
    a column the window covers but the row does not own reaches y -- exactly as on the CPU."""
    m, n = 6, 16
    rp, ci, v = np.array([0, 2, 3, 3, 5, 6, 8], np.int32), np.array([0, 2, 9, 1, 4, 15, 3, 12], np.int32), np.arange(1.0, 9.0)
    x = np.arange(1.0, 17.0)
    x[1] = np.nan  # inside row 0's window [0, 8), not one of its columns
    for rows in (1, 2, 4):
        so, brp, bc, bv, mk = oracle.csr2blkcsr(m, n, 0, rp, ci, v, rows)
        so, yr = oracle.dblkcsrmv(0, 1.0, m, mk, bv, bc, brp, x, 0.0, np.zeros(m), rows)
        st, y = _blkmv(0, 1.0, 0.0, m, n, len(v), mk, bv, bc, brp, x, np.zeros(m), rows, True)
        assert st == 0 and np.isnan(yr[0]) and np.array_equal(y, yr, equal_nan=True)


def test_ellt_all_rows_short_equals_csr_scalar_order():
    """L100-like input: ELLT with one lane per row is the reference's scalar chain, so it must reproduce the CSR
    scalar kernel bit for bit (padding contributes +0)."""
    m, rp, ci, v = laplace5(100)
    w, tc, tv = oracle.csr2ell("ellt", m, 0, rp, ci, v)
    x = np.sin(0.01 * np.arange(m))
    a, b, y = np.array([1.0]), np.array([0.0]), np.zeros(m)
    assert L.aoclsparse_delltmv(P.OP_NONE, P._ptr(a), m, m, len(v), P._ptr(tv), P._ptr(tc), w, P.Descr().h, P._ptr(x), P._ptr(b),
                                P._ptr(y)) == 0
    so, yr = oracle.dcsrmv(0, 0, 1.0, m, len(v), v, ci, rp, x, 0.0, np.zeros(m))
    assert np.array_equal(y, yr)


# --------------------------------------------------------------------------------------------------
# SELL-64: the format optimize builds for an mv hint
# --------------------------------------------------------------------------------------------------
def _hinted(base, m, n, rp, ci, v, kid=None, op=None):
    A = P.Matrix(base, m, n, rp, ci, v)
    d = P.Descr(base=base)
    o = P.OP_NONE if op is None else op
    if kid is None:
        assert L.aoclsparse_set_mv_hint(A.h, o, d.h, 100) == 0
    else:
        assert L.aoclsparse_set_mv_hint_kid(A.h, o, d.h, 100, kid) == 0
    assert L.aoclsparse_optimize(A.h) == 0
    return A, d


def test_sell_chosen_for_uniform_rows_and_bit_exact_scalar_order():
    g = 300
    m, rp, ci, v = laplace5(g)
    A, d = _hinted(0, m, m, rp, ci, v)
    assert A.spmv_info().kernel in (3, 4)
    rng = np.random.default_rng(61)
    x, y0 = rng.uniform(-1, 1, m), rng.uniform(-1, 1, m)
    for alpha, beta in ((1.0, 0.0), (5.1, 3.2)):
        st, y = run_dmv(A, d, x, y0 if beta else np.full(m, np.nan), alpha, beta)
        so, yr = oracle.dcsrmv(-1, 0, alpha, m, len(v), v, ci, rp, x, beta, y0 if beta else np.zeros(m))
        assert st == 0 and np.array_equal(y, yr)
    # NaN / Inf in x only reach the rows that reference them (padding cells are skipped, not multiplied)
    xb = x.copy()
    xb[7], xb[g * 17 + 3] = np.nan, np.inf
    st, y = run_dmv(A, d, xb, np.zeros(m), 1.0, 0.0)
    so, yr = oracle.dcsrmv(-1, 0, 1.0, m, len(v), v, ci, rp, xb, 0.0, np.zeros(m))
    assert np.array_equal(np.isnan(y), np.isnan(yr)) and np.array_equal(y[~np.isnan(yr)], yr[~np.isnan(yr)])


def _mesh(seed, nodes, width, dofs):
    from test_gpu_trsv_blocks import node_mesh
    return node_mesh(seed, nodes, width, dofs, keep=1.0)  # every coupling kept: row lengths uniform enough for SELL


@pytest.mark.parametrize("kid,order", [(None, None), (1, "lane4"), (3, "lane8"), (0, "ref")])
@pytest.mark.parametrize("which", ["five", "four", "mixed"])
def test_sell_shared_column_lists_bit_exact(which, kid, order):
    """mesh matrices (several dofs per node: the rows of a node repeat one column list) are stored as SELL-64 with ONE
    column list per run of rows that share it (kernel 4): groups that straddle a slice boundary, nodes of 2 / 5 / 7 dofs,
    a partial last slice, every summation order, base 1, alpha / beta, NaN / Inf in x, the transposed operator"""
    nodes = 2311
    rng = np.random.default_rng(70)
    dofs = {"five": np.full(nodes, 5), "four": np.full(nodes, 4),
            "mixed": np.repeat([2, 5, 7], [770, 770, 771])}[which]  # three regions: lists of 2, 5 and 7 rows
    m, rp, ci, v = _mesh(71, nodes, 43, dofs)
    assert len(v) >= 8 * m
    x, y0 = rng.uniform(-1, 1, m), rng.uniform(-1, 1, m)
    for base in (0, 1):
        A, d = _hinted(base, m, m, rp + base, ci + base, v, kid=kid)
        assert A.spmv_info().kernel == 4, (which, A.spmv_info().kernel)
        for alpha, beta in ((1.0, 0.0), (-1.3, 0.7)):
            st, y = run_dmv(A, d, x, y0, alpha, beta)
            if order is None:
                so, yr = oracle.dcsrmv(-1, 0, alpha, m, len(v), v, ci, rp, x, beta, y0)
            else:
                so, yr = oracle.dcsrmv_order(order, 0, alpha, m, v, ci, rp, x, beta, y0)
            assert st == 0 and so == 0 and np.array_equal(y, yr), (which, base, kid, alpha)
    xb = x.copy()
    xb[11], xb[m // 2] = np.nan, -np.inf
    st, y = run_dmv(A, d, xb, np.zeros(m), 1.0, 0.0)
    if order is None:
        so, yr = oracle.dcsrmv(-1, 0, 1.0, m, len(v), v, ci, rp, xb, 0.0, np.zeros(m))
    else:
        so, yr = oracle.dcsrmv_order(order, 0, 1.0, m, v, ci, rp, xb, 0.0, np.zeros(m))
    assert np.array_equal(np.isnan(y), np.isnan(yr)) and np.array_equal(y[~np.isnan(yr)], yr[~np.isnan(yr)])


def test_sell_shared_lists_first_entry_differs_base1():
    """rows that repeat the previous row's list everywhere but in their FIRST entry must not be taken for followers -- with
    the index base forgotten in the leader pass (base 1) exactly that entry was never compared"""
    rng = np.random.default_rng(75)
    m, n, L = 64 * 40, 5000, 12
    tail = np.sort(rng.choice(np.arange(100, n), size=L - 1, replace=False))
    rows = [np.concatenate([[int(rng.integers(0, 100))], tail]) for _ in range(m)]  # same tail, own first entry
    rp = (np.arange(m + 1) * L).astype(np.int32)
    ci = np.concatenate(rows).astype(np.int32)
    v = rng.uniform(-1, 1, len(ci))
    x, y0 = rng.uniform(-1, 1, n), rng.uniform(-1, 1, m)
    for base in (0, 1):
        A, d = _hinted(base, m, n, rp + base, ci + base, v)
        assert A.spmv_info().kernel in (3, 4)
        st, y = run_dmv(A, d, x, y0, 1.0, 0.0)
        so, yr = oracle.dcsrmv(-1, 0, 1.0, m, len(v), v, ci, rp, x, 0.0, y0)
        assert st == 0 and so == 0 and np.array_equal(y, yr), base


def test_sell_shared_lists_float_and_transposed():
    nodes = 1500
    m, rp, ci, v = _mesh(72, nodes, 31, np.full(nodes, 4))
    vf = v.astype(np.float32)
    Af = P.Matrix(0, m, m, rp, ci, vf)
    d = P.Descr()
    assert L.aoclsparse_set_mv_hint(Af.h, P.OP_NONE, d.h, 100) == 0 and L.aoclsparse_optimize(Af.h) == 0
    assert Af.spmv_info().kernel == 4
    rng = np.random.default_rng(73)
    xf = rng.uniform(-1, 1, m).astype(np.float32)
    yf = torch.zeros(m, dtype=torch.float32, device="cuda")
    assert P.smv(P.OP_NONE, 1.0, Af, d, dev(xf), 0.0, yf) == 0
    torch.cuda.synchronize()
    # float: the reference runs its 8-lane kernel when nnz > 10 m, the scalar chain otherwise (csrmv.hpp:326-343)
    so, yr = oracle.scsrmv("lane8" if len(v) > 10 * m else "ref", 0, 1.0, m, vf, ci, rp, xf, 0.0, np.zeros(m, np.float32))
    assert so == 0 and np.array_equal(yf.cpu().numpy(), yr)
    # the transposed operator gets its own SELL copy (the mesh pattern is symmetric, so its lists are shared as well)
    B, d2 = _hinted(0, m, m, rp, ci, v, op=P.OP_TRANSPOSE)
    x, y0 = rng.uniform(-1, 1, m), rng.uniform(-1, 1, m)
    st, y = run_dmv(B, d2, x, y0, 0.9, -0.4, op=P.OP_TRANSPOSE)
    so, cp, ri, cv = oracle.dcsr2csc(m, m, len(v), 0, 0, rp, ci, v)
    so, yr = oracle.dcsrmv(-1, 0, 0.9, m, len(v), cv, ri, cp, x, -0.4, y0)
    assert st == 0 and so == 0 and B.spmv_info(P.OP_TRANSPOSE).kernel in (3, 4)
    assert np.max(np.abs(y - yr)) <= 64 * EPS64 * np.max(np.abs(yr))


@pytest.mark.parametrize("kid,order", [(None, "lane8"), (1, "lane4"), (3, "lane8"), (0, "ref")])
@pytest.mark.parametrize("base", [0, 1])
def test_sell_lane_orders_bit_exact(kid, order, base):
    """nnz > 10 m: the reference runs its vector kernels; a SELL lane reproduces their order with 4 / 8 private
    partial sums.  Rows of every length mod 8, empty rows, a last partial slice."""
    m, n = 64 * 37 + 13, 3000
    rng = np.random.default_rng(62)
    lens = rng.integers(36, 41, m)  # padding stays under the budget (1.35 cells per non-zero)
    lens[3], lens[100], lens[m - 1] = 0, 7, 25
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    ci = np.concatenate([rng.choice(n, k, replace=False) for k in lens]).astype(np.int32)  # unsorted columns
    v = rng.uniform(-1, 1, len(ci))
    A, d = _hinted(base, m, n, rp + base, ci + base, v, kid=kid)
    assert A.spmv_info().kernel in (3, 4) and len(v) > 10 * m
    x, y0 = rng.uniform(-1, 1, n), rng.uniform(-1, 1, m)
    st, y = run_dmv(A, d, x, y0, -1.3, 0.7)
    so, yr = oracle.dcsrmv_order(order, 0, -1.3, m, v, ci, rp, x, 0.7, y0)
    assert st == 0 and np.array_equal(y, yr)


def test_sell_float_transposed_and_value_refresh():
    m, n = 2100, 1900
    rp, ci, v = random_csr(63, m, n, lambda r, i: r.integers(12, 15))
    vf = v.astype(np.float32)
    A = P.Matrix(0, m, n, rp, ci, vf)
    d = P.Descr()
    assert L.aoclsparse_set_mv_hint(A.h, P.OP_NONE, d.h, 10) == 0 and L.aoclsparse_optimize(A.h) == 0
    assert A.spmv_info().kernel in (3, 4)
    x = np.random.default_rng(64).uniform(-1, 1, n).astype(np.float32)
    y, a, b = np.zeros(m, np.float32), np.array([1.5], np.float32), np.array([0.0], np.float32)
    assert L.aoclsparse_smv(P.OP_NONE, P._ptr(a), A.h, d.h, P._ptr(x), P._ptr(b), P._ptr(y)) == 0
    so, yr = oracle.scsrmv("lane8", 0, 1.5, m, vf, ci, rp, x, 0.0, np.zeros(m, np.float32))
    assert np.array_equal(y, yr)
    # transposed hint: SELL of A^T (a structurally symmetric pattern keeps the rows of A^T uniform)
    m2, rp2, ci2, v2 = laplace5(50)
    vd = v2 * np.random.default_rng(69).uniform(0.5, 1.5, len(v2))
    B = P.Matrix(0, m2, m2, rp2, ci2, vd)
    assert L.aoclsparse_set_mv_hint(B.h, P.OP_TRANSPOSE, d.h, 10) == 0
    assert L.aoclsparse_set_mv_hint(B.h, P.OP_NONE, d.h, 10) == 0 and L.aoclsparse_optimize(B.h) == 0
    assert B.spmv_info(P.OP_TRANSPOSE).kernel in (3, 4) and B.spmv_info().kernel in (3, 4)
    xt = np.random.default_rng(65).uniform(-1, 1, m2)
    st, yt = run_dmv(B, d, xt, np.zeros(m2), 1.0, 0.0, op=P.OP_TRANSPOSE)
    dense = np.zeros((m2, m2))
    for i in range(m2):
        dense[i, ci2[rp2[i]:rp2[i + 1]]] = vd[rp2[i]:rp2[i + 1]]
    assert st == 0 and np.allclose(yt, dense.T @ xt, rtol=0, atol=1e-12)
    # values change -> the SELL copy is rebuilt on the next product
    nv = np.random.default_rng(66).uniform(-1, 1, len(vd))
    assert L.aoclsparse_dupdate_values(B.h, len(nv), P._ptr(nv)) == 0
    assert B.spmv_info().kernel == 1  # dropped with the other derived copies ...
    xs = np.random.default_rng(67).uniform(-1, 1, m2)
    st, y2 = run_dmv(B, d, xs, np.zeros(m2), 1.0, 0.0)
    so, yr = oracle.dcsrmv(-1, 0, 1.0, m2, len(nv), nv, ci2, rp2, xs, 0.0, np.zeros(m2))
    assert st == 0 and np.array_equal(y2, yr) and B.spmv_info().kernel in (3, 4)  # ... and back on first use


def test_sell_not_chosen_for_power_law_rows():
    m = 20000
    rp, ci, v = random_csr(68, m, m, powerlaw_rows(6, 9000))
    A, d = _hinted(0, m, m, rp, ci, v)
    assert A.spmv_info().kernel in (1, 2)  # a CSR kernel: CSR-Adaptive, or merge-path when the longest row spans >= 16 tiles


def test_user_stream_ordering():
    """aoclsparse_mi355_set_stream: with device operands every executor is enqueued on the caller's HIP stream, ordered with
    the caller's own work on it (a producer kernel before, a consumer after), and nothing runs on the null stream."""
    m, rp, ci, v = laplace5(400)
    A, d = _hinted(0, m, m, rp, ci, v)
    x_h = np.random.default_rng(2).uniform(-1, 1, m)
    so, yr = oracle.dcsrmv(-1, 0, 2.0, m, len(v), v, ci, rp, 3.0 * x_h, 0.0, np.zeros(m))
    stream = torch.cuda.Stream()
    old = L.aoclsparse_mi355_get_stream()
    try:
        assert L.aoclsparse_mi355_set_stream(ctypes.c_void_p(stream.cuda_stream)) == 0
        L.aoclsparse_mi355_set_pointer_mode(P.PTR_DEVICE)
        with torch.cuda.stream(stream):
            xd = dev(x_h)
            big = torch.ones(64 * 1024 * 1024, device="cuda")  # keep the stream busy ahead of the product
            for _ in range(20):
                big.mul_(1.0000001)
            xd.mul_(3.0)  # producer on the same stream: the product must see 3 x
            yd = torch.zeros(m, dtype=torch.float64, device="cuda")
            assert P.dmv(P.OP_NONE, 2.0, A, d, xd, 0.0, yd) == 0
            z = yd * 1.0  # consumer on the same stream
        stream.synchronize()
        assert np.array_equal(z.cpu().numpy(), yr)
    finally:
        L.aoclsparse_mi355_set_stream(ctypes.c_void_p(old))
        L.aoclsparse_mi355_set_pointer_mode(P.PTR_AUTO)


# --------------------------------------------------------------------------------------------------
# the reference's own example programs, compiled unchanged against this library (oracle/Makefile: samples)
# --------------------------------------------------------------------------------------------------
def test_reference_samples_run_unchanged():
    """tests/examples/sample_*.c(pp) of the reference (spmv, csrmm, dotmv, symgs(_mv), trsm, trsv, CG / GMRES direct and RCI
    in both precisions, csr2m, complex sp2m, symgs(_mv), trsm and gthr, axpyi, dotp, roti, sctr, sp2md, spmmd, and the three C++-interface samples through include/aoclsparse.hpp) are built where the reference tree exists and travel as binaries;
    each checks its own result and must exit 0.  Skipped when they were not built."""
    import glob
    import subprocess
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bins = sorted(glob.glob(os.path.join(here, "oracle", "_ref", "samples", "sample_*")))
    if not bins:
        pytest.skip("oracle/_ref/samples not built (no reference tree at build time)")
    for b in bins:
        r = subprocess.run([b], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, (os.path.basename(b), r.stdout[-400:], r.stderr[-400:])
    out = subprocess.run([os.path.join(here, "oracle", "_ref", "samples", "sample_spmv_c")], capture_output=True, text=True).stdout
    assert "69.000000" in out and "40.000000" in out  # y of the sample's 5x5 product


# --------------------------------------------------------------------------------------------------
# randomised sweep: many small shapes through every general-N SpMV route, TRSV and csrmm
# --------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("seed", list(range(24)))
def test_randomised_shapes_spmv_trsv_csrmm(seed):
    """shapes no hand-written case targets: m not a multiple of 64 / 4, empty matrices' neighbours (1 row, 1 column),
    rows of 0..200 entries, both bases, hinted / un-hinted / promoted handles, raw arrays, kid pins, alpha / beta specials.
    SpMV and TRSV bit-exact against the oracle, csrmm bit-exact against the column-major oracle."""
    rng = np.random.default_rng(1000 + seed)
    base = int(rng.integers(0, 2))
    m = int(rng.choice([1, 2, 3, 63, 64, 65, 127, 200, 257, 1000, 2049, 4097]))
    n = int(rng.choice([1, 2, 5, 64, 130, 999, 3000])) if seed % 3 else m
    dens = rng.choice([0, 1, 3, 12, 40, 200])
    rp, ci, v = random_csr(seed, m, n, lambda r, i: 0 if dens == 0 else min(n, int(r.integers(0, 2 * dens + 1))), base=base)
    nnz = len(v)
    if nnz == 0:
        ci, v = np.zeros(1, np.int32) + base, np.zeros(1)
    x, y0 = rng.uniform(-1, 1, n), rng.uniform(-1, 1, m)
    d = P.Descr(base=base)
    for alpha, beta in ((1.0, 0.0), (0.0, 1.0), (-1.5, 0.25), (1.0, 1.0)):
        y0b = np.full(m, np.nan) if beta == 0.0 else y0
        for kid in (-1, 0, 1, 3):
            so, yr = oracle.dcsrmv(kid, base, alpha, m, nnz, v, ci, rp, x, beta, y0b)
            # handle with hint (SELL when the padding allows), pinned kid where given
            A = P.Matrix(base, m, n, rp, ci, v) if nnz else None
            if A is not None:
                if kid < 0:
                    assert L.aoclsparse_set_mv_hint(A.h, P.OP_NONE, d.h, 10) == 0
                else:
                    assert L.aoclsparse_set_mv_hint_kid(A.h, P.OP_NONE, d.h, 10, kid) == 0
                assert L.aoclsparse_optimize(A.h) == 0
                st, y = run_dmv(A, d, x, y0b, alpha, beta)
                lens = np.diff(rp)
                exact = kid >= 0 or lens.max(initial=0) <= A.spmv_info().tile or A.spmv_info().kernel in (3, 4)
                assert st == 0
                if exact:
                    assert np.array_equal(y, yr, equal_nan=True), (kid, alpha, beta)
                else:
                    assert np.allclose(y, yr, rtol=0, atol=1e-12)
            if kid < 0:
                # raw arrays (host pointers) and an un-hinted handle multiplied past its promotion
                yh = y0b.copy()
                st = P.dcsrmv(P.OP_NONE, alpha, m, n, nnz, v, ci, rp, d, x, beta, yh)
                assert st == 0
                if nnz and np.diff(rp).max(initial=0) <= 512:
                    assert np.array_equal(yh, yr, equal_nan=True)
                if nnz:
                    Bu = P.Matrix(base, m, n, rp, ci, v)
                    for _ in range(9):
                        st, y = run_dmv(Bu, d, x, y0b, alpha, beta)
                    assert st == 0 and (np.array_equal(y, yr, equal_nan=True) or np.diff(rp).max() > 512)
    # csrmm against the column-major oracle, both layouts, a few widths
    if nnz:
        A = P.Matrix(base, m, n, rp, ci, v)
        for ncol in (1, 7, 32, 130):
            ldb, ldc = n + int(rng.integers(0, 3)), m + int(rng.integers(0, 3))
            Bc, C0 = rng.uniform(-1, 1, ldb * ncol), rng.uniform(-1, 1, ldc * ncol)
            so, Cr = oracle.dcsrmm("col", -0.5, base, v, ci, rp, m, Bc, ncol, ldb, 2.0, C0, ldc)
            Cd = dev(C0)
            assert P.dcsrmm(P.OP_NONE, -0.5, A, d, P.ORDER_COLUMN, dev(Bc), ncol, ldb, 2.0, Cd, ldc) == 0
            torch.cuda.synchronize()
            assert np.array_equal(Cd.cpu().numpy(), Cr), ("col", ncol)
            ldb2, ldc2 = ncol + int(rng.integers(0, 3)), ncol + int(rng.integers(0, 3))
            Br = np.zeros((n, ldb2)); Br[:, :ncol] = Bc.reshape(ncol, ldb)[:, :n].T
            Cr0 = np.full((m, ldc2), 7.25); Cr0[:, :ncol] = C0.reshape(ncol, ldc)[:, :m].T
            Cd = dev(Cr0.ravel())
            assert P.dcsrmm(P.OP_NONE, -0.5, A, d, P.ORDER_ROW, dev(Br.ravel()), ncol, ldb2, 2.0, Cd, ldc2) == 0
            torch.cuda.synchronize()
            got = Cd.cpu().numpy().reshape(m, ldc2)
            assert np.array_equal(got[:, :ncol], Cr.reshape(ncol, ldc)[:, :m].T) and np.all(got[:, ncol:] == 7.25), ("row", ncol)
    # TRSV on a square system of the same size class
    if m >= 2:
        trp, tci, tv = triangular_system(seed, m, int(rng.integers(1, 6)), base=base)
        T = P.Matrix(base, m, m, trp, tci, tv)
        b = rng.uniform(-1, 1, m)
        o = oracle.dcsr_optimize(m, m, len(tv), base, trp, tci, tv)
        for fill, kind, ends in ((P.FILL_LOWER, "l", "idiag"), (P.FILL_UPPER, "u", "iurow")):
            for unit in (False, True):
                dt = P.Descr(base=base, mtype=P.TYPE_TRIANGULAR, fill=fill, diag=P.DIAG_UNIT if unit else P.DIAG_NON_UNIT)
                for kid in (-1, 0, 1, 3):
                    lanes = kt_lanes(kid)
                    if lanes:
                        so, xr = oracle.trsv_kt(kind, lanes, 0.75, m, o["base"], o["val"], o["ind"], o["ptr"], o[ends], b, unit)
                    else:
                        so, xr = oracle.dtrsv(kind, 0.75, m, o["base"], o["val"], o["ind"], o["ptr"], o[ends], b, unit)
                    xd = dev(np.zeros(m))
                    assert P.dtrsv(P.OP_NONE, 0.75, T, dt, dev(b), xd, kid=kid) == 0
                    torch.cuda.synchronize()
                    assert np.array_equal(xd.cpu().numpy(), xr[:m]), (kind, unit, kid)


@pytest.mark.parametrize("seed", list(range(12)))
def test_randomised_shapes_transposes_float_sp2m_trsm(seed):
    """second sweep: transposed / strided TRSV and trsm (bit-exact), float SpMV in its 8-lane order (bit-exact), transposed
    and symmetric dmv (forward-error bound), sp2m on rectangular shapes down to one row (structure and values bit-exact)."""
    rng = np.random.default_rng(2000 + seed)
    base = seed % 2
    m = int(rng.choice([1, 3, 64, 65, 300, 1025, 3000]))
    n = int(rng.choice([1, 4, 70, 500, 2500]))
    rp, ci, v = random_csr(100 + seed, m, n, lambda r, i: min(n, int(r.integers(0, 25))), base=base)
    nnz, d = len(v), P.Descr(base=base)
    if nnz:
        # float, general N: the reference's 8-lane order
        vf = v.astype(np.float32)
        xf, yf0 = rng.uniform(-1, 1, n).astype(np.float32), rng.uniform(-1, 1, m).astype(np.float32)
        Af = P.Matrix(base, m, n, rp, ci, vf)
        so, yr = oracle.scsrmv("lane8", base, 0.5, m, vf, ci, rp, xf, -1.0, yf0)
        yd = dev(yf0)
        assert P.smv(P.OP_NONE, 0.5, Af, d, dev(xf), -1.0, yd) == 0
        torch.cuda.synchronize()
        assert np.array_equal(yd.cpu().numpy(), yr)
        # double, transposed: bound against a dense product
        A = P.Matrix(base, m, n, rp, ci, v)
        D = _dense(m, n, rp, ci, v, base)
        xt, yt0 = rng.uniform(-1, 1, m), rng.uniform(-1, 1, n)
        st, yt = run_dmv(A, d, xt, yt0, 1.25, -0.5, op=P.OP_TRANSPOSE)
        scale = 1.25 * (np.abs(D).T @ np.abs(xt)) + 0.5 * np.abs(yt0)
        assert st == 0 and np.all(np.abs(yt - (1.25 * D.T @ xt - 0.5 * yt0)) <= (np.count_nonzero(D, axis=0) + 6) * EPS64 * scale + 1e-300)
        # sp2m: A * B with B = n x k
        k = int(rng.choice([1, 9, 400]))
        pb, ib, vb = random_csr(300 + seed, n, k, lambda r, i: min(k, int(r.integers(0, 12))), base=1 - base)
        if len(vb):
            Bm = P.Matrix(1 - base, n, k, pb, ib, vb)
            C = ctypes.c_void_p()
            assert L.aoclsparse_sp2m(P.OP_NONE, d.h, A.h, P.OP_NONE, P.Descr(base=1 - base).h, Bm.h, P.STAGE_FULL, ctypes.byref(C)) == 0
            _, cm, cn, cz, row, col, val = _export(C)
            so, pc, ic, vc = oracle.dcsr2m(m, k, base, rp, ci, v, 1 - base, pb, ib, vb)
            assert (cm, cn) == (m, k) and np.array_equal(row, pc) and np.array_equal(col, ic) and np.array_equal(val, vc)
            L.aoclsparse_destroy(ctypes.byref(C))
    # triangular solves: transposed, strided, and several right-hand sides
    mt = int(rng.choice([2, 63, 130, 1500]))
    trp, tci, tv = triangular_system(500 + seed, mt, int(rng.integers(1, 5)), base=base)
    T = P.Matrix(base, mt, mt, trp, tci, tv)
    b = rng.uniform(-1, 1, mt)
    for fill in ("lower", "upper"):
        for unit in (False, True):
            dt = P.Descr(base=base, mtype=P.TYPE_TRIANGULAR, fill=P.FILL_LOWER if fill == "lower" else P.FILL_UPPER,
                         diag=P.DIAG_UNIT if unit else P.DIAG_NON_UNIT)
            xr = oracle_trsv(base, mt, trp, tci, tv, fill, "t", unit, -0.6, b)
            for kid in (None, 0, 3):
                x = np.zeros(mt)
                assert P.dtrsv(P.OP_TRANSPOSE, -0.6, T, dt, b, x, kid=kid) == 0 and np.array_equal(x, xr[:mt])
            bs, xs = np.zeros(2 * mt), np.full(3 * mt, 5.5)
            bs[::2] = b
            assert P.dtrsv(P.OP_TRANSPOSE, -0.6, T, dt, bs, xs, incb=2, incx=3) == 0
            assert np.array_equal(xs[::3], xr[:mt]) and np.all(xs[1::3] == 5.5) and np.all(xs[2::3] == 5.5)
            nr = int(rng.choice([1, 2, 9]))
            Bm, Xm = rng.uniform(-1, 1, (mt, nr)), np.zeros((mt, nr))
            assert L.aoclsparse_dtrsm(P.OP_NONE, 2.0, T.h, dt.h, P.ORDER_ROW, P._ptr(Bm), nr, nr, P._ptr(Xm), nr) == 0
            for j in range(nr):
                assert np.array_equal(Xm[:, j], oracle_trsv(base, mt, trp, tci, tv, fill, "n", unit, 2.0, Bm[:, j].copy())[:mt])


@pytest.mark.parametrize("seed", list(range(10)))
def test_randomised_shapes_complex(seed):
    """third sweep, complex handles: zmv / cmv (general, N / T / H), zcsrmm (both layouts), ztrsv (L / U, N / T / H), zdotmv and
    sp2m on random shapes down to 1 x 1, both bases, against dense numpy operators within forward-error bounds."""
    rng = np.random.default_rng(3000 + seed)
    base = seed % 2
    m = int(rng.choice([1, 2, 63, 65, 300, 1100]))
    n = int(rng.choice([1, 3, 64, 257, 900]))
    rp, ci, vr = random_csr(700 + seed, m, n, lambda r, i: min(n, int(r.integers(0, 14))), base=base)
    if len(vr) == 0:
        return
    for dtype, eps, pre in ((np.complex128, EPS64, "z"), (np.complex64, EPS32, "c")):
        fn = lambda stem: getattr(L, "aoclsparse_" + stem.replace("?", pre))
        C = P.CDouble if pre == "z" else P.CFloat
        v = (vr + 1j * rng.uniform(-1, 1, len(vr))).astype(dtype)
        D = np.zeros((m, n), np.complex128)
        for i in range(m):
            D[i, ci[rp[i] - base:rp[i + 1] - base] - base] = v[rp[i] - base:rp[i + 1] - base]
        h = ctypes.c_void_p()
        assert fn("create_?csr")(ctypes.byref(h), base, m, n, len(v), P._ptr(rp), P._ptr(ci), P._ptr(v)) == 0
        d = P.Descr(base=base)
        alpha, beta = np.array([0.7 - 0.2j], dtype), np.array([-0.4 + 0.3j], dtype)
        for op, M in ((P.OP_NONE, D), (P.OP_TRANSPOSE, D.T), (P.OP_CONJ_TRANSPOSE, D.conj().T)):
            x = (rng.uniform(-1, 1, M.shape[1]) + 1j * rng.uniform(-1, 1, M.shape[1])).astype(dtype)
            y0 = (rng.uniform(-1, 1, M.shape[0]) + 1j * rng.uniform(-1, 1, M.shape[0])).astype(dtype)
            y = y0.copy()
            assert fn("?mv")(op, P._ptr(alpha), h, d.h, P._ptr(x), P._ptr(beta), P._ptr(y)) == 0
            ref = alpha[0] * (M @ x.astype(np.complex128)) + beta[0] * y0
            scale = abs(alpha[0]) * (np.abs(M) @ np.abs(x)) + abs(beta[0]) * np.abs(y0)
            assert np.all(np.abs(y - ref) <= (2 * np.count_nonzero(M, axis=1) + 16) * eps * scale + 1e-30), (pre, op)
            # csrmm, 5 columns, both layouts
            k = 5
            Bm = (rng.uniform(-1, 1, (M.shape[1], k)) + 1j * rng.uniform(-1, 1, (M.shape[1], k))).astype(dtype)
            C0 = (rng.uniform(-1, 1, (M.shape[0], k)) + 1j * rng.uniform(-1, 1, (M.shape[0], k))).astype(dtype)
            refm = alpha[0] * (M @ Bm.astype(np.complex128)) + beta[0] * C0
            scm = abs(alpha[0]) * (np.abs(M) @ np.abs(Bm)) + abs(beta[0]) * np.abs(C0)
            for order in (P.ORDER_ROW, P.ORDER_COLUMN):
                Bb = np.ascontiguousarray(Bm if order == P.ORDER_ROW else Bm.T)
                Cb = np.ascontiguousarray(C0 if order == P.ORDER_ROW else C0.T).copy()
                ldb, ldc = (k, k) if order == P.ORDER_ROW else (M.shape[1], M.shape[0])
                assert fn("?csrmm")(op, C(alpha[0].real, alpha[0].imag), h, d.h, order, P._ptr(Bb), k, ldb,
                                    C(beta[0].real, beta[0].imag), P._ptr(Cb), ldc) == 0
                got = Cb if order == P.ORDER_ROW else Cb.T
                assert np.all(np.abs(got - refm) <= (2 * np.count_nonzero(M, axis=1)[:, None] + 80) * eps * scm + 1e-30), (pre, op, order)
        L.aoclsparse_destroy(ctypes.byref(h))
    # triangular solves and dotmv on a square diagonally dominant system
    nt = int(rng.choice([2, 64, 65, 333]))
    dense, trp, tci, tv = _cplx_tri_system(900 + seed, nt, np.complex128, base)
    h = ctypes.c_void_p()
    assert L.aoclsparse_create_zcsr(ctypes.byref(h), base, nt, nt, len(tv), P._ptr(trp), P._ptr(tci), P._ptr(tv)) == 0
    b = rng.uniform(-1, 1, nt) + 1j * rng.uniform(-1, 1, nt)
    for fill in ("lower", "upper"):
        for op, nm in ((P.OP_NONE, "n"), (P.OP_TRANSPOSE, "t"), (P.OP_CONJ_TRANSPOSE, "h")):
            dt = P.Descr(base=base, mtype=P.TYPE_TRIANGULAR, fill=P.FILL_LOWER if fill == "lower" else P.FILL_UPPER)
            xr = np.linalg.solve(_op_tri(dense.astype(np.complex128), fill, "non_unit", nm), (0.5 + 0.5j) * b)
            x = np.zeros(nt, np.complex128)
            assert L.aoclsparse_ztrsv(op, P.CDouble(0.5, 0.5), h, dt.h, P._ptr(b), P._ptr(x)) == 0
            assert np.max(np.abs(x - xr)) <= 64 * EPS64 * max(1.0, np.max(np.abs(xr)))
    dg = P.Descr(base=base)
    y, dot = np.zeros(nt, np.complex128), np.zeros(1, np.complex128)
    assert L.aoclsparse_zdotmv(P.OP_NONE, P.CDouble(1, 0), h, dg.h, P._ptr(b), P.CDouble(0, 0), P._ptr(y), P._ptr(dot)) == 0
    yr = dense.astype(np.complex128) @ b
    assert np.max(np.abs(y - yr)) <= 64 * EPS64 * np.max(np.abs(yr)) and abs(dot[0] - np.vdot(b, yr)) <= 4 * nt * EPS64 * np.sum(np.abs(b) * np.abs(yr))
    L.aoclsparse_destroy(ctypes.byref(h))


# --------------------------------------------------------------------------------------------------
# iterative solvers
# --------------------------------------------------------------------------------------------------

def _sym_full_test(n, rp, ci, v):
    rows = [[] for _ in range(n)]
    for i in range(n):
        for p in range(rp[i], rp[i + 1]):
            rows[i].append((ci[p], v[p]))
            if ci[p] != i:
                rows[ci[p]].append((i, v[p]))
    rows = [sorted(r) for r in rows]
    frp = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int32)
    return frp, np.array([c for r in rows for c, _ in r], np.int32), np.array([a for r in rows for _, a in r], np.float64)


def torch_from_ptr(ptr, n):
    """float64 CUDA tensor aliasing n elements of device memory at `ptr` (RCI workspaces handed out by the library)."""
    class _Holder:
        pass
    hld = _Holder()
    hld.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 2}
    return torch.as_tensor(hld, device="cuda")


PRECOND_T = ctypes.CFUNCTYPE(ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(ctypes.c_double),
                             ctypes.POINTER(ctypes.c_double), ctypes.c_void_p)
MONIT_T = ctypes.CFUNCTYPE(ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
                           ctypes.POINTER(ctypes.c_double), ctypes.c_void_p)


def _itsol(opts):
    h = ctypes.c_void_p()
    assert L.aoclsparse_itsol_d_init(ctypes.byref(h)) == 0
    for k, v in opts.items():
        assert L.aoclsparse_itsol_option_set(h, k.encode(), str(v).encode()) == 0, (k, v)
    return h


def _lower(n, rp, ci, v):
    keep = np.concatenate([[p for p in range(rp[i], rp[i + 1]) if ci[p] <= i] for i in range(n)]).astype(np.int64)
    lens = [np.count_nonzero(ci[rp[i]:rp[i + 1]] <= i) for i in range(n)]
    return np.concatenate([[0], np.cumsum(lens)]).astype(np.int32), ci[keep].copy(), v[keep].copy()


def test_itsol_cg_reference_example(kats):
    """tests/examples/sample_itsol_d_cg.cpp: lower-stored 8x8 SPD system, SGS-preconditioned CG, host vectors."""
    c = kats["itsol"]["cg"]
    n = c["n"]
    rp, ci, v = np.array(c["row_ptr"], np.int32), np.array(c["col_ind"], np.int32), np.array(c["val"], np.float64)
    A = P.Matrix(0, n, n, rp, ci, v)
    d = P.Descr(mtype=P.TYPE_SYMMETRIC, fill=P.FILL_LOWER)
    xe, b = np.array(c["expected"]), np.zeros(n)
    assert P.dmv(P.OP_NONE, 1.0, A, d, xe, 0.0, b) == 0
    for pre in ("None", "SGS"):
        h = _itsol({"CG Abs Tolerance": c["abs_tol"], "CG Preconditioner": pre})
        x, rinfo = np.array(c["x0"]), np.zeros(100)
        assert L.aoclsparse_itsol_d_solve(h, n, A.h, d.h, P._ptr(b), P._ptr(x), P._ptr(rinfo), None, None, None) == 0
        assert np.max(np.abs(x - xe)) < 1e-5 and rinfo[0] <= c["abs_tol"] and 1 <= rinfo[30] <= 8
        assert abs(rinfo[1] - np.linalg.norm(b)) <= 1e-12 * np.linalg.norm(b)
        L.aoclsparse_itsol_destroy(ctypes.byref(h))
        assert h.value is None


@pytest.mark.parametrize("pre,code", [("None", 0), ("SymGS", 3)])
def test_itsol_cg_laplacian_matches_restated_solver(pre, code):
    """Same iteration count (+-1: different dot-product trees) and solution as the restated CPU solver; device
    vectors stay in HBM for the whole solve."""
    g = 64
    n, rp, ci, v = laplace5(g)
    lrp, lci, lv = _lower(n, rp, ci, v)
    A = P.Matrix(0, n, n, lrp, lci, lv)
    d = P.Descr(mtype=P.TYPE_SYMMETRIC, fill=P.FILL_LOWER)
    rng = np.random.default_rng(71)
    xe = rng.uniform(-1, 1, n)
    so, b = oracle.dcsrmv(0, 0, 1.0, n, len(v), v, ci, rp, xe, 0.0, np.zeros(n))
    o = oracle.dcsr_optimize(n, n, len(v), 0, rp, ci, v)
    st, xo, ro = oracle.dcg(n, 0, o["ptr"], o["ind"], o["val"], o["idiag"], o["iurow"], b, np.zeros(n), 1e-9, 0.0, 500, code)
    assert st == 0
    h = _itsol({"CG Rel Tolerance": 1e-9, "CG Abs Tolerance": 0.0, "CG Preconditioner": pre, "CG Iteration Limit": 500})
    xd, bd, rinfo = dev(np.zeros(n)), dev(b), np.zeros(100)
    assert L.aoclsparse_itsol_d_solve(h, n, A.h, d.h, P._ptr(bd), P._ptr(xd), P._ptr(rinfo), None, None, None) == 0
    x = xd.cpu().numpy()
    assert abs(rinfo[30] - ro[30]) <= 1, (rinfo[30], ro[30])
    assert np.max(np.abs(x - xe)) < 1e-6 and np.max(np.abs(x - xo)) < 1e-6
    assert rinfo[0] <= 1e-9 * np.linalg.norm(b)
    L.aoclsparse_itsol_destroy(ctypes.byref(h))


def test_itsol_gmres_ilu0_and_plain():
    """Unsymmetric, diagonally dominant system: restarted GMRES with and without the ILU(0) preconditioner against
    the restated solver (iterations counted per completed restart cycle, as the reference does)."""
    g = 40
    n, rp, ci, v = laplace5(g)
    v = v.copy()
    rng = np.random.default_rng(72)
    v[v < 0] = rng.uniform(-1.0, -0.3, np.count_nonzero(v < 0))
    xe = rng.uniform(-1, 1, n)
    so, b = oracle.dcsrmv(0, 0, 1.0, n, len(v), v, ci, rp, xe, 0.0, np.zeros(n))
    A = P.Matrix(0, n, n, rp, ci, v)
    d = P.Descr()
    for pre, code in (("None", 0), ("ILU0", 2)):
        st, xo, ro = oracle.dgmres(n, 0, rp, ci, v, b, np.ones(n), 20, 1e-10, 1e-12, 400, code)
        h = _itsol({"iterative method": "GMRES", "gmres preconditioner": pre, "gmres restart iterations": 20,
                    "gmres rel tolerance": 1e-10, "gmres abs tolerance": 1e-12, "gmres iteration limit": 400})
        x, rinfo = np.ones(n), np.zeros(100)
        sg = L.aoclsparse_itsol_d_solve(h, n, A.h, d.h, P._ptr(b), P._ptr(x), P._ptr(rinfo), None, None, None)
        assert sg == st == 0, (pre, sg, st, rinfo[0], rinfo[30])
        assert abs(rinfo[30] - ro[30]) <= 20 and np.max(np.abs(x - xe)) < 1e-7, (pre, rinfo[30], ro[30])
        L.aoclsparse_itsol_destroy(ctypes.byref(h))


def test_itsol_callbacks_limits_and_errors(kats):
    c = kats["itsol"]["cg"]
    n = c["n"]
    rp, ci, v = np.array(c["row_ptr"], np.int32), np.array(c["col_ind"], np.int32), np.array(c["val"], np.float64)
    A = P.Matrix(0, n, n, rp, ci, v)
    d = P.Descr(mtype=P.TYPE_SYMMETRIC, fill=P.FILL_LOWER)
    xe, b = np.array(c["expected"]), np.zeros(n)
    assert P.dmv(P.OP_NONE, 1.0, A, d, xe, 0.0, b) == 0
    diag = np.array([19, 10, 11, 13, 11, 9, 12, 9], np.float64)
    calls = {"p": 0, "m": 0}

    def jacobi(flag, nn, u, w, udata):
        calls["p"] += 1
        for i in range(nn):
            w[i] = u[i] / diag[i]
        return 0

    def monit(nn, x, r, rinfo, udata):
        calls["m"] += 1
        assert abs(np.linalg.norm([r[i] for i in range(nn)]) - rinfo[0]) <= 1e-10 * max(1.0, rinfo[0])
        return 0

    h = _itsol({"CG Preconditioner": "User", "CG Abs Tolerance": 1e-9})
    x, rinfo = np.ones(n), np.zeros(100)
    pc, mc = PRECOND_T(jacobi), MONIT_T(monit)
    assert L.aoclsparse_itsol_d_solve(h, n, A.h, d.h, P._ptr(b), P._ptr(x), P._ptr(rinfo), pc, mc, None) == 0
    assert np.max(np.abs(x - xe)) < 1e-7 and calls["p"] == rinfo[30] and calls["m"] >= rinfo[30]
    # user stop from the monitor (status 8), missing user preconditioner (2), iteration limit (7)
    stop = MONIT_T(lambda nn, x, r, ri, u: 1 if ri[30] > 1 else 0)
    x = np.ones(n)
    assert L.aoclsparse_itsol_d_solve(h, n, A.h, d.h, P._ptr(b), P._ptr(x), P._ptr(rinfo), pc, stop, None) == 8
    assert L.aoclsparse_itsol_d_solve(h, n, A.h, d.h, P._ptr(b), P._ptr(x), P._ptr(rinfo), None, None, None) == 2
    L.aoclsparse_itsol_destroy(ctypes.byref(h))
    h = _itsol({"CG Iteration Limit": 2, "CG Abs Tolerance": 1e-14, "CG Rel Tolerance": 0.0})
    x = np.ones(n)
    assert L.aoclsparse_itsol_d_solve(h, n, A.h, d.h, P._ptr(b), P._ptr(x), P._ptr(rinfo), None, None, None) == 7
    assert rinfo[30] == 3  # the reference stops once niter > maxit
    assert L.aoclsparse_itsol_d_solve(h, n, A.h, P.Descr().h, P._ptr(b), P._ptr(x), P._ptr(rinfo), None, None, None) == 5
    assert L.aoclsparse_itsol_d_solve(h, n + 1, A.h, d.h, P._ptr(np.zeros(n + 1)), P._ptr(np.zeros(n + 1)), P._ptr(rinfo), None,
                                      None, None) == 3
    assert L.aoclsparse_itsol_s_solve(h, n, A.h, d.h, P._ptr(b), P._ptr(x), P._ptr(rinfo), None, None, None) == 9
    L.aoclsparse_itsol_destroy(ctypes.byref(h))


@pytest.mark.parametrize("device", [False, True])
def test_itsol_rci_interfaces(device, kats):
    """Reverse communication: the caller runs v = A u itself.  Host b -> pinned workspaces a host caller can read;
    device b -> HBM workspaces and device pointers."""
    c = kats["itsol"]["cg"]
    n = c["n"]
    rp, ci, v = _sym_full_test(n, c["row_ptr"], c["col_ind"], c["val"])
    dense = np.zeros((n, n))
    for i in range(n):
        dense[i, ci[rp[i]:rp[i + 1]]] = v[rp[i]:rp[i + 1]]
    xe = np.array(c["expected"])
    b = dense @ xe
    h = _itsol({"CG Abs Tolerance": 1e-10})
    rinfo, ircomm = np.zeros(100), ctypes.c_int(1)
    u, w = ctypes.c_void_p(), ctypes.c_void_p()
    if device:
        bd, xd = dev(b), dev(np.ones(n))
        assert L.aoclsparse_itsol_d_rci_input(h, n, P._ptr(bd)) == 0
        xarg = P._ptr(xd)
    else:
        x = np.ones(n)
        assert L.aoclsparse_itsol_d_rci_input(h, n, P._ptr(b)) == 0
        xarg = P._ptr(x)
    Ad = dev(dense.reshape(-1))
    steps = 0
    while ircomm.value != 0 and steps < 200:
        st = L.aoclsparse_itsol_d_rci_solve(h, ctypes.byref(ircomm), ctypes.byref(u), ctypes.byref(w), xarg, P._ptr(rinfo))
        assert st == 0
        steps += 1
        if ircomm.value == 2:  # aoclsparse_rci_mv
            if device:
                assert L.aoclsparse_mi355_synchronize() == 0
                uu = torch_from_ptr(u.value, n)
                ww = torch_from_ptr(w.value, n)
                ww.copy_(Ad.view(n, n) @ uu)
                torch.cuda.synchronize()
            else:
                uu = np.ctypeslib.as_array(ctypes.cast(u, ctypes.POINTER(ctypes.c_double)), (n,))
                ww = np.ctypeslib.as_array(ctypes.cast(w, ctypes.POINTER(ctypes.c_double)), (n,))
                ww[:] = dense @ uu
    assert ircomm.value == 0 and steps < 200
    xs = xd.cpu().numpy() if device else x
    assert np.max(np.abs(xs - xe)) < 1e-8 and rinfo[0] <= 1e-10
    L.aoclsparse_itsol_destroy(ctypes.byref(h))


def test_ilu_factorisation_on_gpu_long_rows_partial_sort_and_bad_pivot():
    """The level-scheduled factorisation equals the serial IKJ loop bit for bit also for rows longer than a
    wavefront, for partially sorted rows (L | D | U groups, unsorted inside), and reports a vanishing pivot."""
    rng = np.random.default_rng(81)
    n = 600
    dense = np.zeros((n, n))
    for i in range(n):
        cols = np.unique(np.clip(i + rng.integers(-150, 151, 90), 0, n - 1))
        dense[i, cols] = rng.uniform(-1, 1, len(cols))
        dense[i, i] = 40.0 + rng.uniform(0, 1)
    rows = []
    for i in range(n):
        c = np.flatnonzero(dense[i])
        lo, up = c[c < i], c[c > i]
        rng.shuffle(lo), rng.shuffle(up)  # partially sorted: groups in order, unsorted inside
        rows.append(np.concatenate([lo, [i], up]))
    rp = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int32)
    ci = np.concatenate(rows).astype(np.int32)
    v = np.array([dense[i, c] for i, r in enumerate(rows) for c in r])
    assert np.diff(rp).max() > 64
    A = P.Matrix(0, n, n, rp, ci, v)
    d = P.Descr()
    st, lu, diag = oracle.dilu0(n, 0, rp, ci, v)
    assert st == 0
    pv, b, x = ctypes.c_void_p(), rng.uniform(-1, 1, n), np.zeros(n)
    assert L.aoclsparse_dilu_smoother(P.OP_NONE, A.h, d.h, ctypes.byref(pv), None, P._ptr(x), P._ptr(b)) == 0
    fac = np.ctypeslib.as_array(ctypes.cast(pv, ctypes.POINTER(ctypes.c_double)), (len(v),))
    assert np.array_equal(fac, lu)
    st, xr = oracle.dilu_solve(n, 0, diag, lu, rp, ci, b)
    assert np.array_equal(x, xr)
    # a pivot that cancels exactly: [[1, 1], [1, 1]] -> u_11 = 0 -> numerical_error (11)
    rp2, ci2, v2 = np.array([0, 2, 4], np.int32), np.array([0, 1, 0, 1], np.int32), np.array([1.0, 1.0, 1.0, 1.0])
    B = P.Matrix(0, 2, 2, rp2, ci2, v2)
    assert L.aoclsparse_dilu_smoother(P.OP_NONE, B.h, d.h, ctypes.byref(pv), None, P._ptr(np.zeros(2)), P._ptr(np.ones(2))) == 11
    assert oracle.dilu0(2, 0, rp2, ci2, v2)[0] != 0


def test_csc_handle_and_converted_coo_run_the_path():
    """A handle created from CSC arrays must behave as the CSR handle of the same matrix in dmv (N / T), trsv and
    csrmm; a COO handle converted with aoclsparse_convert_csr likewise."""
    m = 900
    rp, ci, v = triangular_system(101, m, 6, band=60)
    so, cp, ri, cv = oracle.dcsr2csc(m, m, len(v), 0, 0, rp, ci, v)
    hc = ctypes.c_void_p()
    assert L.aoclsparse_create_dcsc(ctypes.byref(hc), 0, m, m, len(v), P._ptr(cp), P._ptr(ri), P._ptr(cv)) == 0
    rng = np.random.default_rng(102)
    x, one, zero = rng.uniform(-1, 1, m), np.array([1.0]), np.array([0.0])
    d = P.Descr()
    for op, orc in ((P.OP_NONE, lambda: oracle.dcsrmv(-1, 0, 1.0, m, len(v), v, ci, rp, x, 0.0, np.zeros(m))[1]),
                    (P.OP_TRANSPOSE, lambda: oracle.dcsrmvt(0, 1.0, m, m, v, ci, rp, x, 0.0, np.zeros(m))[1])):
        y = np.zeros(m)
        assert L.aoclsparse_dmv(op, P._ptr(one), hc, d.h, P._ptr(x), P._ptr(zero), P._ptr(y)) == 0
        yr = orc()
        assert np.array_equal(y, yr) if op == P.OP_NONE else np.allclose(y, yr, rtol=0, atol=1e-12)
    dt = P.Descr(mtype=P.TYPE_TRIANGULAR, fill=P.FILL_LOWER)
    b, xs = rng.uniform(-1, 1, m), np.zeros(m)
    assert L.aoclsparse_dtrsv(P.OP_NONE, 1.5, hc, dt.h, P._ptr(b), P._ptr(xs)) == 0
    assert np.array_equal(xs, oracle_trsv(0, m, rp, ci, v, "lower", "n", False, 1.5, b))
    L.aoclsparse_destroy(ctypes.byref(hc))
    # COO (shuffled) -> CSR -> order_mat -> same product as the original CSR
    perm = rng.permutation(len(v))
    cr = np.repeat(np.arange(m, dtype=np.int32), np.diff(rp))[perm].copy()
    cc, cval = ci[perm].copy(), v[perm].copy()
    hcoo, hcsr = ctypes.c_void_p(), ctypes.c_void_p()
    assert L.aoclsparse_create_dcoo(ctypes.byref(hcoo), 0, m, m, len(v), P._ptr(cr), P._ptr(cc), P._ptr(cval)) == 0
    assert L.aoclsparse_convert_csr(hcoo, P.OP_NONE, ctypes.byref(hcsr)) == 0 and L.aoclsparse_order_mat(hcsr) == 0
    y = np.zeros(m)
    assert L.aoclsparse_dmv(P.OP_NONE, P._ptr(one), hcsr, d.h, P._ptr(x), P._ptr(zero), P._ptr(y)) == 0
    so, yr = oracle.dcsrmv(-1, 0, 1.0, m, len(v), v, ci, rp, x, 0.0, np.zeros(m))
    assert np.array_equal(y, yr)
    L.aoclsparse_destroy(ctypes.byref(hcsr)), L.aoclsparse_destroy(ctypes.byref(hcoo))


# --------------------------------------------------------------------------------------------------
# complex handles
# --------------------------------------------------------------------------------------------------
def _cplx_matrix(seed, m, n, maxlen, dtype):
    rng = np.random.default_rng(seed)
    lens = rng.integers(0, maxlen, m)
    lens[min(3, m - 1)] = 0
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    ci = np.concatenate([np.sort(rng.choice(n, k, replace=False)) for k in lens] + [np.zeros(0, np.int64)]).astype(np.int32)
    v = (rng.uniform(-1, 1, len(ci)) + 1j * rng.uniform(-1, 1, len(ci))).astype(dtype)
    return rp, ci, v


@pytest.mark.parametrize("prec", ["z", "c"])
def test_complex_mv_every_descriptor_and_operation(prec):
    """aoclsparse_zmv / aoclsparse_cmv: general (rectangular), symmetric, hermitian, triangular x N / T / H x fill x
    diag, both index bases, host and device vectors, within (len + 8) eps of the restated operator."""
    dtype, rdtype, eps = (np.complex128, np.float64, EPS64) if prec == "z" else (np.complex64, np.float32, EPS32)
    create = L.aoclsparse_create_zcsr if prec == "z" else L.aoclsparse_create_ccsr
    mv = L.aoclsparse_zmv if prec == "z" else L.aoclsparse_cmv
    rng = np.random.default_rng(111)
    alpha, beta = np.array([0.7 - 0.4j], dtype), np.array([-0.3 + 0.2j], dtype)
    ops = {"n": P.OP_NONE, "t": P.OP_TRANSPOSE, "h": P.OP_CONJ_TRANSPOSE}
    for base in (0, 1):
        for (m, n, cases) in ((700, 640, [("general", "lower", "non_unit")]),
                              (600, 600, [(t, f, dg) for t in ("symmetric", "hermitian", "triangular") for f in ("lower", "upper")
                                          for dg in (("non_unit", "unit") if f == "lower" else ("non_unit", "zero"))])):
            rp, ci, v = _cplx_matrix(112 + base, m, n, 19, dtype)
            if m == n:  # full diagonal so that every diag type is meaningful
                dense = np.zeros((m, n), dtype)
                for i in range(m):
                    dense[i, ci[rp[i]:rp[i + 1]]] = v[rp[i]:rp[i + 1]]
                    dense[i, i] = 3.0  # real: a hermitian matrix has a real diagonal
                rows = [np.flatnonzero(dense[i]) for i in range(m)]
                rp = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int32)
                ci = np.concatenate(rows).astype(np.int32)
                v = np.concatenate([dense[i, r] for i, r in enumerate(rows)]).astype(dtype)
            rpb, cib = rp + base, ci + base
            h = ctypes.c_void_p()
            assert create(ctypes.byref(h), base, m, n, len(v), P._ptr(rpb), P._ptr(cib), P._ptr(v)) == 0
            lens = np.diff(rp)
            for mtype, fill, diag in cases:
                d = P.Descr(base=base, mtype={"general": 0, "symmetric": 1, "hermitian": 2, "triangular": 3}[mtype],
                            fill=P.FILL_LOWER if fill == "lower" else P.FILL_UPPER,
                            diag={"non_unit": 0, "unit": 1, "zero": 2}[diag])
                for opn, op in ops.items():
                    nx, ny = (n, m) if opn == "n" or mtype != "general" else (m, n)
                    x = (rng.uniform(-1, 1, nx) + 1j * rng.uniform(-1, 1, nx)).astype(dtype)
                    y0 = (rng.uniform(-1, 1, ny) + 1j * rng.uniform(-1, 1, ny)).astype(dtype)
                    yr, scale = oracle.zmv(opn, mtype, fill, diag, base, alpha[0], m, n, rpb, cib, v, x, beta[0], y0)
                    y = y0.copy()
                    assert mv(op, P._ptr(alpha), h, d.h, P._ptr(x), P._ptr(beta), P._ptr(y)) == 0, (mtype, opn)
                    bound = (2 * lens.max() + 16) * eps * (scale + 1e-30)
                    assert np.all(np.abs(y - yr) <= bound), (prec, base, mtype, fill, diag, opn, np.max(np.abs(y - yr) / bound))
                    yd = dev(y0)
                    assert mv(op, P._ptr(alpha), h, d.h, P._ptr(dev(x)), P._ptr(beta), P._ptr(yd)) == 0
                    torch.cuda.synchronize()
                    assert np.array_equal(yd.cpu().numpy(), y)
            # beta = 0 must not read y
            zero, x = np.zeros(1, dtype), (rng.uniform(-1, 1, n)).astype(dtype)
            y = np.full(m, np.nan + 1j * np.nan, dtype)
            assert mv(P.OP_NONE, P._ptr(alpha), h, P.Descr(base=base).h, P._ptr(x), P._ptr(zero), P._ptr(y)) == 0
            assert not np.any(np.isnan(y))
            L.aoclsparse_destroy(ctypes.byref(h))


@pytest.mark.parametrize("prec", ["z", "c"])
def test_complex_dotmv(prec):
    """aoclsparse_{c,z}dotmv: y as ?mv (same kernel, same bits), d = sum conj(x_i) y_i over min(m, n) entries
    (dense_dot.hpp:36-49) within 2k eps sum|x_i y_i| of numpy's vdot; host and device operands."""
    dtype, eps = (np.complex128, EPS64) if prec == "z" else (np.complex64, EPS32)
    C = P.CDouble if prec == "z" else P.CFloat
    create = L.aoclsparse_create_zcsr if prec == "z" else L.aoclsparse_create_ccsr
    mv, dotmv = (L.aoclsparse_zmv, L.aoclsparse_zdotmv) if prec == "z" else (L.aoclsparse_cmv, L.aoclsparse_cdotmv)
    m, n = 900, 700
    rp, ci, v = _cplx_matrix(131, m, n, 9, dtype)
    h = ctypes.c_void_p()
    assert create(ctypes.byref(h), 0, m, n, len(v), P._ptr(rp), P._ptr(ci), P._ptr(v)) == 0
    d = P.Descr()
    rng = np.random.default_rng(9)
    alpha, beta = np.array([0.6 + 0.3j], dtype), np.array([-0.2 + 0.7j], dtype)
    for op, nx, ny in ((P.OP_NONE, n, m), (P.OP_CONJ_TRANSPOSE, m, n)):
        x = (rng.uniform(-1, 1, nx) + 1j * rng.uniform(-1, 1, nx)).astype(dtype)
        y0 = (rng.uniform(-1, 1, ny) + 1j * rng.uniform(-1, 1, ny)).astype(dtype)
        yr = y0.copy()
        assert mv(op, P._ptr(alpha), h, d.h, P._ptr(x), P._ptr(beta), P._ptr(yr)) == 0
        y, dot = y0.copy(), np.zeros(1, dtype)
        assert dotmv(op, C(alpha[0].real, alpha[0].imag), h, d.h, P._ptr(x), C(beta[0].real, beta[0].imag), P._ptr(y), P._ptr(dot)) == 0
        k = min(m, n)
        ref = np.vdot(x[:k].astype(np.complex128), yr[:k].astype(np.complex128))
        assert np.array_equal(y, yr)
        assert abs(dot[0] - ref) <= 2 * k * eps * np.sum(np.abs(x[:k]) * np.abs(yr[:k]))
        yd, dd = dev(y0), dev(np.zeros(1, dtype))
        assert dotmv(op, C(alpha[0].real, alpha[0].imag), h, d.h, P._ptr(dev(x)), C(beta[0].real, beta[0].imag), P._ptr(yd), P._ptr(dd)) == 0
        torch.cuda.synchronize()
        assert np.array_equal(yd.cpu().numpy(), yr) and dd.cpu().numpy()[0] == dot[0]
    assert dotmv(P.OP_NONE, C(1, 0), h, d.h, P._ptr(x), C(0, 0), P._ptr(y), None) == 2
    L.aoclsparse_destroy(ctypes.byref(h))


def test_complex_handle_plumbing_and_type_checks():
    rp, ci, v = _cplx_matrix(113, 40, 40, 6, np.complex128)
    h = ctypes.c_void_p()
    assert L.aoclsparse_create_zcsr(ctypes.byref(h), 0, 40, 40, len(v), P._ptr(rp), P._ptr(ci), P._ptr(v)) == 0
    b_, m_, n_, z_ = ctypes.c_int(), ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
    a1, a2, a3 = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
    outs = (ctypes.byref(b_), ctypes.byref(m_), ctypes.byref(n_), ctypes.byref(z_), ctypes.byref(a1), ctypes.byref(a2), ctypes.byref(a3))
    assert L.aoclsparse_export_zcsr(h, *outs) == 0 and a3.value == v.ctypes.data and z_.value == len(v)
    assert L.aoclsparse_export_dcsr(h, *outs) == 9 and L.aoclsparse_export_ccsr(h, *outs) == 9
    r = int(np.flatnonzero(np.diff(rp))[0])
    assert L.aoclsparse_zset_value(h, r, int(ci[rp[r]]), P.CDouble(2.5, -1.5)) == 0 and v[rp[r]] == 2.5 - 1.5j
    nv = (np.arange(len(v)) + 1j).astype(np.complex128)
    assert L.aoclsparse_zupdate_values(h, len(v), P._ptr(nv)) == 0 and np.array_equal(v, nv)
    d, one = P.Descr(), np.ones(1)
    x, y = np.ones(40), np.zeros(40)
    assert L.aoclsparse_dmv(P.OP_NONE, P._ptr(one), h, d.h, P._ptr(x), P._ptr(one), P._ptr(y)) == 9  # wrong_type
    assert L.aoclsparse_cmv(P.OP_NONE, P._ptr(one), h, d.h, P._ptr(x), P._ptr(one), P._ptr(y)) == 9
    assert L.aoclsparse_set_mv_hint(h, P.OP_NONE, d.h, 5) == 0 and L.aoclsparse_optimize(h) == 0
    c = ctypes.c_void_p()
    assert L.aoclsparse_copy(h, d.h, ctypes.byref(c)) == 0
    xz, yz, a = (np.ones(40) + 0j), np.zeros(40, np.complex128), np.ones(1, np.complex128)
    y2 = yz.copy()
    assert L.aoclsparse_zmv(P.OP_NONE, P._ptr(a), h, d.h, P._ptr(xz), P._ptr(np.zeros(1, np.complex128)), P._ptr(yz)) == 0
    assert L.aoclsparse_zmv(P.OP_NONE, P._ptr(a), c, d.h, P._ptr(xz), P._ptr(np.zeros(1, np.complex128)), P._ptr(y2)) == 0
    assert np.array_equal(yz, y2) and np.allclose(yz, np.add.reduceat(nv, rp[:-1].clip(max=len(nv) - 1)) * (np.diff(rp) > 0))
    L.aoclsparse_destroy(ctypes.byref(c)), L.aoclsparse_destroy(ctypes.byref(h))


def test_csrsv_bit_exact():
    """aoclsparse_?csrsv on sorted rows with a full diagonal: the chain of csrsv.hpp:88-187 bit for bit (lower / upper,
    unit / non-unit, alpha), float within its own restated chain via the double check of structure."""
    m = 3000
    rp, ci, v = triangular_system(81, m, 6)
    b = np.random.default_rng(4).uniform(-1, 1, m)
    for lower in (True, False):
        for unit in (False, True):
            d = P.Descr(fill=P.FILL_LOWER if lower else P.FILL_UPPER, diag=P.DIAG_UNIT if unit else P.DIAG_NON_UNIT)
            so, yr = oracle.dcsrsv(lower, unit, -1.3, m, v, ci, rp, b)
            y, a = np.full(m, np.nan), np.array([-1.3])
            assert L.aoclsparse_dcsrsv(P.OP_NONE, P._ptr(a), m, P._ptr(v), P._ptr(ci), P._ptr(rp), d.h, P._ptr(b), P._ptr(y)) == 0
            assert np.array_equal(y, yr)
    vf, bf, yf, af = v.astype(np.float32), b.astype(np.float32), np.zeros(m, np.float32), np.array([1.0], np.float32)
    assert L.aoclsparse_scsrsv(P.OP_NONE, P._ptr(af), m, P._ptr(vf), P._ptr(ci), P._ptr(rp), P.Descr().h, P._ptr(bf), P._ptr(yf)) == 0
    so, yr = oracle.dcsrsv(True, False, 1.0, m, vf.astype(np.float64), ci, rp, bf.astype(np.float64))
    assert np.max(np.abs(yf - yr)) <= 64 * EPS32 * np.max(np.abs(yr))
    # a missing diagonal in a non-unit solve is refused (the reference would divide by a stale entry)
    rp2, ci2, v2 = np.array([0, 1, 2], np.int32), np.array([0, 0], np.int32), np.array([2.0, 1.0])
    y2 = np.zeros(2)
    assert L.aoclsparse_dcsrsv(P.OP_NONE, P._ptr(np.array([1.0])), 2, P._ptr(v2), P._ptr(ci2), P._ptr(rp2), P.Descr().h,
                               P._ptr(np.ones(2)), P._ptr(y2)) == 5


@pytest.mark.parametrize("prec", ["z", "c"])
def test_complex_sp2m(prec):
    """aoclsparse_sp2m / aoclsparse_spmm on complex handles: the structure (row_ptr and the first-touch column order) is the
    one the real product of the same patterns has -- bit-exact against the oracle -- and the values are within
    (terms + 4) eps sum|a||b| of the dense product; op in {N, T, H} on either side conjugates as it transposes
    (csr2m.cpp:743-835); one row long enough for the global-slab path; two-stage protocol."""
    dtype, eps = (np.complex128, EPS64) if prec == "z" else (np.complex64, EPS32)
    create = L.aoclsparse_create_zcsr if prec == "z" else L.aoclsparse_create_ccsr
    exp = L.aoclsparse_export_zcsr if prec == "z" else L.aoclsparse_export_ccsr
    rng = np.random.default_rng(77)

    def mat(seed, m, n, rl):
        rp, ci, vr = random_csr(seed, m, n, rl)
        v = (vr + 1j * rng.uniform(-1, 1, len(vr))).astype(dtype)
        h = ctypes.c_void_p()
        assert create(ctypes.byref(h), 0, m, n, len(v), P._ptr(rp), P._ptr(ci), P._ptr(v)) == 0
        D = np.zeros((m, n), np.complex128)
        for i in range(m):
            D[i, ci[rp[i]:rp[i + 1]]] = v[rp[i]:rp[i + 1]]
        return h, (rp, ci, v, vr), D

    def result(C):
        b, m_, n_, z_ = ctypes.c_int(), ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
        a1, a2, a3 = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        assert exp(C, ctypes.byref(b), ctypes.byref(m_), ctypes.byref(n_), ctypes.byref(z_), ctypes.byref(a1), ctypes.byref(a2), ctypes.byref(a3)) == 0
        nz = z_.value
        row = np.ctypeslib.as_array(ctypes.cast(a1, ctypes.POINTER(ctypes.c_int32)), (m_.value + 1,)).copy()
        col = np.ctypeslib.as_array(ctypes.cast(a2, ctypes.POINTER(ctypes.c_int32)), (max(nz, 1),))[:nz].copy()
        val = np.ctypeslib.as_array(ctypes.cast(a3, ctypes.POINTER(ctypes.c_float if prec == "c" else ctypes.c_double)), (2 * max(nz, 1),)).view(dtype)[:nz].copy()
        D = np.zeros((m_.value, n_.value), np.complex128)
        for i in range(m_.value):
            assert len(set(col[row[i]:row[i + 1]])) == row[i + 1] - row[i]
            D[i, col[row[i]:row[i + 1]]] = val[row[i]:row[i + 1]]
        return b.value, row, col, D

    m, k, n = 300, 260, 280
    hA, (pa, ia, va, var), DA = mat(1, m, k, lambda r, i: 400 if i == 9 else r.integers(0, 8))
    hB, (pb, ib, vb, vbr), DB = mat(2, k, n, lambda r, i: r.integers(0, 9))
    d = P.Descr()
    opv = {"n": P.OP_NONE, "t": P.OP_TRANSPOSE, "h": P.OP_CONJ_TRANSPOSE}
    f = {"n": lambda X: X, "t": lambda X: X.T, "h": lambda X: X.conj().T}
    # N * N: structure against the oracle on the real parts (same patterns), values against the dense product
    C = ctypes.c_void_p()
    assert L.aoclsparse_sp2m(P.OP_NONE, d.h, hA, P.OP_NONE, d.h, hB, P.STAGE_FULL, ctypes.byref(C)) == 0
    b, row, col, D = result(C)
    so, pc, ic, vc = oracle.dcsr2m(m, n, 0, pa, ia, var, 0, pb, ib, vbr)
    assert b == 0 and np.array_equal(row, pc) and np.array_equal(col, ic)
    bound = (np.abs(DA) @ np.abs(DB)) * (12 * eps) + 1e-300
    assert np.all(np.abs(D - DA @ DB) <= bound)
    L.aoclsparse_destroy(ctypes.byref(C))
    # two-stage protocol gives the same handle and result
    assert L.aoclsparse_sp2m(P.OP_NONE, d.h, hA, P.OP_NONE, d.h, hB, P.STAGE_NNZ_COUNT, ctypes.byref(C)) == 0
    h0 = C.value
    assert L.aoclsparse_sp2m(P.OP_NONE, d.h, hA, P.OP_NONE, d.h, hB, P.STAGE_FINALIZE, ctypes.byref(C)) == 0 and C.value == h0
    assert np.array_equal(result(C)[3], D)
    L.aoclsparse_destroy(ctypes.byref(C))
    # every operation pair on conforming shapes
    shapes = {"n": (m, k), "t": (k, m), "h": (k, m)}
    for oa in "nth":
        for ob in "nth":
            ra, ca_ = shapes[oa]
            h1, keep1, D1 = mat(10 + ord(oa), ra, ca_, lambda r, i: r.integers(0, 7))  # the handle aliases keep1's arrays
            inner = k  # columns of op(A1) in every case
            rb, cb = (inner, 90) if ob == "n" else (90, inner)
            h2, keep2, D2 = mat(20 + ord(ob), rb, cb, lambda r, i: r.integers(0, 7))
            assert L.aoclsparse_sp2m(opv[oa], d.h, h1, opv[ob], d.h, h2, P.STAGE_FULL, ctypes.byref(C)) == 0
            _, _, _, Dc = result(C)
            ref = f[oa](D1) @ f[ob](D2)
            bound = (np.abs(f[oa](D1)) @ np.abs(f[ob](D2))) * (12 * eps) + 1e-300
            assert Dc.shape == ref.shape and np.all(np.abs(Dc - ref) <= bound), (oa, ob)
            L.aoclsparse_destroy(ctypes.byref(C)), L.aoclsparse_destroy(ctypes.byref(h1)), L.aoclsparse_destroy(ctypes.byref(h2))
    # mixed value types are refused
    hr = P.Matrix(0, k, n, pb, ib, vbr)
    assert L.aoclsparse_sp2m(P.OP_NONE, d.h, hA, P.OP_NONE, d.h, hr.h, P.STAGE_FULL, ctypes.byref(C)) == 9
    L.aoclsparse_destroy(ctypes.byref(hA)), L.aoclsparse_destroy(ctypes.byref(hB))


def _cplx_tri_system(seed, n, dtype, base):
    """sorted complex CSR with a dominant full diagonal, ~8 entries per row on both sides of it"""
    rng = np.random.default_rng(seed)
    dense = np.zeros((n, n), np.complex128)
    for i in range(n):
        cols = rng.choice(n, size=min(n, 8), replace=False)
        dense[i, cols] = rng.uniform(-0.5, 0.5, len(cols)) + 1j * rng.uniform(-0.5, 0.5, len(cols))
        dense[i, i] = (3.0 + rng.uniform(0, 1)) * np.exp(1j * rng.uniform(0, 2 * np.pi))
    dense = dense.astype(dtype)
    rows = [np.flatnonzero(dense[i]) for i in range(n)]
    rp = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int32) + base
    ci = (np.concatenate(rows) + base).astype(np.int32)
    v = np.concatenate([dense[i, r] for i, r in enumerate(rows)]).astype(dtype)
    return dense, rp, ci, v


def _op_tri(dense, fill, diag, op):
    T = np.tril(dense) if fill == "lower" else np.triu(dense)
    if diag == "unit":
        np.fill_diagonal(T, 1.0)
    return {"n": T, "t": T.T, "h": T.conj().T}[op]


@pytest.mark.parametrize("prec", ["z", "c"])
def test_complex_trsv_trsm(prec):
    """aoclsparse_{c,z}trsv(_kid)(_strided) and {c,z}trsm: L / U x N / T / H x unit / non-unit, both bases, host and
    device operands, strides and padded leading dimensions, against a dense solve of the same triangle.  The reference's
    complex arithmetic is std::complex (trsv_kr.hpp:38-222), so parity is normwise: 64 eps |x| for these
    diagonally dominant systems."""
    dtype, eps = (np.complex128, EPS64) if prec == "z" else (np.complex64, EPS32)
    C = P.CDouble if prec == "z" else P.CFloat
    fn = lambda stem: getattr(L, "aoclsparse_" + stem.replace("?", prec))
    ops = {"n": P.OP_NONE, "t": P.OP_TRANSPOSE, "h": P.OP_CONJ_TRANSPOSE}
    n = 500
    rng = np.random.default_rng(5)
    alpha = 0.8 - 0.6j
    for base in (0, 1):
        dense, rp, ci, v = _cplx_tri_system(31 + base, n, dtype, base)
        h = ctypes.c_void_p()
        assert fn("create_?csr")(ctypes.byref(h), base, n, n, len(v), P._ptr(rp), P._ptr(ci), P._ptr(v)) == 0
        b = (rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)).astype(dtype)
        for fill in ("lower", "upper"):
            for diag in ("non_unit", "unit"):
                d = P.Descr(base=base, mtype=P.TYPE_TRIANGULAR, fill=P.FILL_LOWER if fill == "lower" else P.FILL_UPPER,
                            diag=P.DIAG_UNIT if diag == "unit" else P.DIAG_NON_UNIT)
                for op in "nth":
                    M = _op_tri(dense.astype(np.complex128), fill, diag, op)
                    xr = np.linalg.solve(M, alpha * b.astype(np.complex128))
                    tol = 64 * eps * np.max(np.abs(xr))
                    x = np.full(n, np.nan, dtype)
                    assert fn("?trsv")(ops[op], C(alpha.real, alpha.imag), h, d.h, P._ptr(b), P._ptr(x)) == 0
                    assert np.max(np.abs(x - xr)) <= tol, (fill, diag, op, np.max(np.abs(x - xr)), tol)
                    xd = dev(np.zeros(n, dtype))
                    assert fn("?trsv_kid")(ops[op], C(alpha.real, alpha.imag), h, d.h, P._ptr(dev(b)), P._ptr(xd), 3) == 0
                    torch.cuda.synchronize()
                    assert np.array_equal(xd.cpu().numpy(), x)  # one arithmetic whatever the operands' home
            # strided, and multi-RHS in both layouts with padded leading dimensions (padding untouched)
            d = P.Descr(base=base, mtype=P.TYPE_TRIANGULAR, fill=P.FILL_LOWER if fill == "lower" else P.FILL_UPPER)
            xr = np.linalg.solve(_op_tri(dense.astype(np.complex128), fill, "non_unit", "h"), alpha * b.astype(np.complex128))
            bs, xs = np.zeros(3 * n, dtype), np.full(2 * n, 7 + 7j, dtype)
            bs[::3] = b
            assert fn("?trsv_strided")(P.OP_CONJ_TRANSPOSE, C(alpha.real, alpha.imag), h, d.h, P._ptr(bs), 3, P._ptr(xs), 2) == 0
            assert np.max(np.abs(xs[::2] - xr)) <= 64 * eps * np.max(np.abs(xr)) and np.all(xs[1::2] == 7 + 7j)
            k = 5
            Bm = (rng.uniform(-1, 1, (n, k)) + 1j * rng.uniform(-1, 1, (n, k))).astype(dtype)
            Xr = np.linalg.solve(_op_tri(dense.astype(np.complex128), fill, "non_unit", "t"), alpha * Bm.astype(np.complex128))
            for order, ldb, ldx in ((P.ORDER_ROW, k + 2, k + 1), (P.ORDER_COLUMN, n + 3, n + 4)):
                if order == P.ORDER_ROW:
                    Bb, Xb = np.zeros((n, ldb), dtype), np.full((n, ldx), 9 - 9j, dtype)
                    Bb[:, :k] = Bm
                else:
                    Bb, Xb = np.zeros((k, ldb), dtype), np.full((k, ldx), 9 - 9j, dtype)
                    Bb[:, :n] = Bm.T
                assert fn("?trsm")(P.OP_TRANSPOSE, C(alpha.real, alpha.imag), h, d.h, order, P._ptr(Bb), k, ldb, P._ptr(Xb), ldx) == 0
                got = Xb[:, :k] if order == P.ORDER_ROW else Xb[:, :n].T
                pad = Xb[:, k:] if order == P.ORDER_ROW else Xb[:, n:]
                assert np.max(np.abs(got - Xr)) <= 64 * eps * np.max(np.abs(Xr)) and np.all(pad == 9 - 9j)
        # checks shared with the real solves (trsv.cpp:59-137)
        dg = P.Descr(base=base)
        assert fn("?trsv")(P.OP_NONE, C(1, 0), h, dg.h, P._ptr(b), P._ptr(x)) == 5  # general descriptor
        wrong = L.aoclsparse_ctrsv if prec == "z" else L.aoclsparse_ztrsv
        assert wrong(P.OP_NONE, (P.CFloat if prec == "z" else P.CDouble)(1, 0), h, d.h, P._ptr(b), P._ptr(x)) == 9
        assert fn("?trsv_kid")(P.OP_NONE, C(1, 0), h, d.h, P._ptr(b), P._ptr(x), 4) == 14
        L.aoclsparse_destroy(ctypes.byref(h))


@pytest.mark.parametrize("prec", ["z", "c"])
def test_complex_symgs(prec):
    """aoclsparse_{c,z}symgs(_mv): the two triangular sweeps of symgs.hpp:62-258 restated with dense numpy operators --
    general (N, T), symmetric and hermitian (both fills): x1 = (L+D)^-1 (b - alpha U x0), x = (U+D)^-1 (b - L x1),
    y = op(A) x; within 256 eps of the restatement on diagonally dominant systems."""
    dtype, eps = (np.complex128, EPS64) if prec == "z" else (np.complex64, EPS32)
    C = P.CDouble if prec == "z" else P.CFloat
    fn = lambda stem: getattr(L, "aoclsparse_" + stem.replace("?", prec))
    n = 300
    rng = np.random.default_rng(21)
    dense, rp, ci, v = _cplx_tri_system(77, n, dtype, 0)
    D = dense.astype(np.complex128)
    h = ctypes.c_void_p()
    assert fn("create_?csr")(ctypes.byref(h), 0, n, n, len(v), P._ptr(rp), P._ptr(ci), P._ptr(v)) == 0
    b = (rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)).astype(dtype)
    x0 = (rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)).astype(dtype)
    alpha = 0.9 + 0.2j
    sl, su, dg = np.tril(D, -1), np.triu(D, 1), np.diag(np.diag(D))
    cases = [("general", "lower", P.OP_NONE, sl, su, D), ("general", "lower", P.OP_TRANSPOSE, su.T, sl.T, D.T),
             ("symmetric", "lower", P.OP_NONE, sl, sl.T, sl + dg + sl.T), ("symmetric", "upper", P.OP_NONE, su.T, su, su + dg + su.T),
             ("hermitian", "lower", P.OP_NONE, sl, sl.conj().T, sl + dg + sl.conj().T),
             ("hermitian", "upper", P.OP_NONE, su.conj().T, su, su + dg + su.conj().T)]
    types = {"general": P.TYPE_GENERAL, "symmetric": P.TYPE_SYMMETRIC, "hermitian": P.TYPE_HERMITIAN}
    for tname, fill, op, Lm, Um, Afull in cases:
        d = P.Descr(mtype=types[tname], fill=P.FILL_LOWER if fill == "lower" else P.FILL_UPPER)
        # the sweep that runs on the conjugate transpose of the stored triangle sees conj(D) (trsv with op = H)
        Dl = dg.conj() if (tname, fill) == ("hermitian", "upper") else dg
        Du = dg.conj() if (tname, fill) == ("hermitian", "lower") else dg
        x1 = np.linalg.solve(Lm + Dl, b.astype(np.complex128) - alpha * (Um @ x0.astype(np.complex128)))
        xr = np.linalg.solve(Um + Du, b.astype(np.complex128) - Lm @ x1)
        x, y = x0.copy(), np.zeros(n, dtype)
        assert fn("?symgs_mv")(op, h, d.h, C(alpha.real, alpha.imag), P._ptr(b), P._ptr(x), P._ptr(y)) == 0
        tol = 256 * eps * max(1.0, np.max(np.abs(xr)))
        assert np.max(np.abs(x - xr)) <= tol, (tname, fill, op, np.max(np.abs(x - xr)), tol)
        yr = Afull @ xr
        assert np.max(np.abs(y - yr)) <= 256 * eps * max(1.0, np.max(np.abs(yr))) * 8, (tname, fill)
        xd = dev(x0)
        assert fn("?symgs")(op, h, d.h, C(alpha.real, alpha.imag), P._ptr(dev(b)), P._ptr(xd)) == 0
        torch.cuda.synchronize()
        assert np.array_equal(xd.cpu().numpy(), x)
    dgen = P.Descr()
    assert fn("?symgs")(P.OP_CONJ_TRANSPOSE, h, dgen.h, C(1, 0), P._ptr(b), P._ptr(x)) == 1  # symgs.hpp: not implemented
    L.aoclsparse_destroy(ctypes.byref(h))


@pytest.mark.parametrize("prec", ["z", "c"])
def test_complex_ilu_smoother(prec):
    """aoclsparse_{c,z}ilu_smoother: ILU(0) on the pattern (ilu0.hpp:34-111, IKJ) restated with numpy complex arithmetic,
    then L y = b (unit lower), U x = y; factors through *precond_csr_val and x within 64 eps of the restatement."""
    dtype, eps = (np.complex128, EPS64) if prec == "z" else (np.complex64, EPS32)
    fn = lambda stem: getattr(L, "aoclsparse_" + stem.replace("?", prec))
    n = 400
    dense, rp, ci, v = _cplx_tri_system(91, n, dtype, 0)
    h = ctypes.c_void_p()
    assert fn("create_?csr")(ctypes.byref(h), 0, n, n, len(v), P._ptr(rp), P._ptr(ci), P._ptr(v)) == 0
    # restated factorisation on the pattern
    lu = v.astype(np.complex128).copy()
    pos = [dict((int(ci[p]), p) for p in range(rp[i], rp[i + 1])) for i in range(n)]
    for i in range(n):
        for p in range(rp[i], rp[i + 1]):
            k = int(ci[p])
            if k >= i:
                break
            lu[p] = lu[p] / lu[pos[k][k]]
            for q in range(pos[k][k] + 1, rp[k + 1]):
                w = pos[i].get(int(ci[q]))
                if w is not None:
                    lu[w] -= lu[p] * lu[q]
    Lm, Um = np.eye(n, dtype=np.complex128), np.zeros((n, n), np.complex128)
    for i in range(n):
        for p in range(rp[i], rp[i + 1]):
            (Lm if ci[p] < i else Um)[i, ci[p]] = lu[p]
    rng = np.random.default_rng(3)
    b = (rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)).astype(dtype)
    xr = np.linalg.solve(Um, np.linalg.solve(Lm, b.astype(np.complex128)))
    d = P.Descr()
    assert L.aoclsparse_set_lu_smoother_hint(h, P.OP_NONE, d.h, 1) == 0 and L.aoclsparse_optimize(h) == 0
    x, pv = np.zeros(n, dtype), ctypes.c_void_p()
    assert fn("?ilu_smoother")(P.OP_NONE, h, d.h, ctypes.byref(pv), None, P._ptr(x), P._ptr(b)) == 0
    got = np.ctypeslib.as_array(ctypes.cast(pv, ctypes.POINTER(ctypes.c_float if prec == "c" else ctypes.c_double)), (2 * len(v),)).view(dtype)
    assert np.max(np.abs(got - lu)) <= 64 * eps * np.max(np.abs(lu))
    assert np.max(np.abs(x - xr)) <= 256 * eps * np.max(np.abs(xr))
    x2 = dev(np.zeros(n, dtype))
    assert fn("?ilu_smoother")(P.OP_NONE, h, d.h, ctypes.byref(pv), None, P._ptr(x2), P._ptr(dev(b))) == 0
    torch.cuda.synchronize()
    assert np.array_equal(x2.cpu().numpy(), x)
    L.aoclsparse_destroy(ctypes.byref(h))


@pytest.mark.parametrize("prec", ["z", "c"])
def test_complex_itsol_cg_and_gmres(prec):
    """aoclsparse_itsol_{c,z}_solve / _rci_*: the reference's state machines over complex vectors.  CG (its unconjugated
    form) on a complex SYMMETRIC diagonally dominant system, GMRES (restarted, also with the ILU(0) preconditioner) on a
    general one; exit status, iteration count (CG +-1, GMRES same cycle) and solution against the numpy restatement, and
    the RCI loop driven by hand gives the direct interface's bits."""
    dtype, rdtype, eps = (np.complex128, np.float64, EPS64) if prec == "z" else (np.complex64, np.float32, EPS32)
    fn = lambda stem: getattr(L, "aoclsparse_" + stem.replace("?", prec))
    n = 200
    rng = np.random.default_rng(8)
    dense, rp, ci, v = _cplx_tri_system(55, n, dtype, 0)
    # a spectrum in the right half plane (the generator's diagonal has random phases: eigenvalues all around the origin)
    for i in range(n):
        dgp = rp[i] + int(np.searchsorted(ci[rp[i]:rp[i + 1]], i))
        v[dgp] = 4.0 + 0.5j * (1 + i % 3)
        dense[i, i] = v[dgp]
    D = dense.astype(np.complex128)
    xs = (rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)).astype(np.complex128)
    tol = 1e-9 if prec == "z" else 2e-4

    def handle(method, extra=()):
        hdl = ctypes.c_void_p()
        assert fn("itsol_?_init")(ctypes.byref(hdl)) == 0
        for k, val in (("iterative method", method),) + tuple(extra):
            assert L.aoclsparse_itsol_option_set(hdl, k.encode(), val.encode()) == 0
        return hdl

    # ---- GMRES on the general matrix
    A = ctypes.c_void_p()
    assert fn("create_?csr")(ctypes.byref(A), 0, n, n, len(v), P._ptr(rp), P._ptr(ci), P._ptr(v)) == 0
    d = P.Descr()
    b = (D @ xs).astype(dtype)
    opts = (("gmres rel tolerance", str(tol)), ("gmres abs tolerance", "0"), ("gmres restart iterations", "15"),
            ("gmres iteration limit", "300"))
    st_r, xr, it_r, rn_r = oracle.zgmres(D, b.astype(np.complex128), np.zeros(n), 15, tol, 0.0, 300)
    hdl = handle("gmres", opts)
    x, rinfo = np.zeros(n, dtype), np.zeros(100, rdtype)
    assert fn("itsol_?_solve")(hdl, n, A, d.h, P._ptr(b), P._ptr(x), P._ptr(rinfo), None, None, None) == st_r == 0
    assert abs(int(rinfo[30]) - it_r) <= 15 and np.max(np.abs(x - xs)) <= 50 * tol * np.max(np.abs(xs))
    # the same solve through the RCI interface, mv done by the caller with aoclsparse_?mv on the handed-out pointers
    hdl2 = handle("gmres", opts)
    assert fn("itsol_?_rci_input")(hdl2, n, P._ptr(b)) == 0
    x2, rinfo2 = np.zeros(n, dtype), np.zeros(100, rdtype)
    job, u, vv = ctypes.c_int(1), ctypes.c_void_p(), ctypes.c_void_p()  # aoclsparse_rci_start
    one, zero = np.ones(1, dtype), np.zeros(1, dtype)
    mv = fn("?mv")
    for _ in range(5000):
        st = fn("itsol_?_rci_solve")(hdl2, ctypes.byref(job), ctypes.byref(u), ctypes.byref(vv), P._ptr(x2), P._ptr(rinfo2))
        assert st == 0
        if job.value == 2:  # aoclsparse_rci_mv
            assert mv(P.OP_NONE, P._ptr(one), A, d.h, u, P._ptr(zero), vv) == 0
        elif job.value == 0:  # aoclsparse_rci_stop
            break
    assert job.value == 0 and np.array_equal(x2, x) and rinfo2[30] == rinfo[30]
    # ILU(0)-preconditioned: converges in no more cycles than the plain solve
    hdl3 = handle("gmres", opts + (("gmres preconditioner", "ilu0"),))
    assert L.aoclsparse_set_lu_smoother_hint(A, P.OP_NONE, d.h, 1) == 0 and L.aoclsparse_optimize(A) == 0
    x3, rinfo3 = np.zeros(n, dtype), np.zeros(100, rdtype)
    assert fn("itsol_?_solve")(hdl3, n, A, d.h, P._ptr(b), P._ptr(x3), P._ptr(rinfo3), None, None, None) == 0
    assert rinfo3[30] <= rinfo[30] and np.max(np.abs(x3 - xs)) <= 50 * tol * np.max(np.abs(xs))
    for hh in (hdl, hdl2, hdl3):
        L.aoclsparse_itsol_destroy(ctypes.byref(hh))
    L.aoclsparse_destroy(ctypes.byref(A))
    # ---- CG on the complex symmetric matrix S = tril + tril^T (lower triangle stored)
    Ls = np.tril(D)
    S = Ls + np.tril(D, -1).T
    rows = [np.flatnonzero(Ls[i]) for i in range(n)]
    lrp = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int32)
    lci = np.concatenate(rows).astype(np.int32)
    lv = np.concatenate([Ls[i, r] for i, r in enumerate(rows)]).astype(dtype)
    A = ctypes.c_void_p()
    assert fn("create_?csr")(ctypes.byref(A), 0, n, n, len(lv), P._ptr(lrp), P._ptr(lci), P._ptr(lv)) == 0
    ds = P.Descr(mtype=P.TYPE_SYMMETRIC, fill=P.FILL_LOWER)
    b = (S @ xs).astype(dtype)
    st_r, xr, it_r, rn_r = oracle.zcg(S, b.astype(np.complex128), np.zeros(n), tol, 0.0, 500)
    hdl = handle("cg", (("cg rel tolerance", str(tol)), ("cg abs tolerance", "0"), ("cg iteration limit", "500")))
    x, rinfo = np.zeros(n, dtype), np.zeros(100, rdtype)
    assert fn("itsol_?_solve")(hdl, n, A, ds.h, P._ptr(b), P._ptr(x), P._ptr(rinfo), None, None, None) == st_r == 0
    assert abs(int(rinfo[30]) - it_r) <= (1 if prec == "z" else 3) and np.max(np.abs(x - xs)) <= 200 * tol * np.max(np.abs(xs))
    # the built-in SymGS preconditioner: fewer iterations, same solution
    hdl4 = handle("cg", (("cg preconditioner", "symgs"), ("cg rel tolerance", str(tol)), ("cg abs tolerance", "0")))
    x4, rinfo4 = np.zeros(n, dtype), np.zeros(100, rdtype)
    assert fn("itsol_?_solve")(hdl4, n, A, ds.h, P._ptr(b), P._ptr(x4), P._ptr(rinfo4), None, None, None) == 0
    assert rinfo4[30] < rinfo[30] and np.max(np.abs(x4 - xs)) <= 200 * tol * np.max(np.abs(xs))
    L.aoclsparse_itsol_destroy(ctypes.byref(hdl)), L.aoclsparse_itsol_destroy(ctypes.byref(hdl4))
    L.aoclsparse_destroy(ctypes.byref(A))


def test_complex_trsv_reference_h5_round_trip():
    """trsv_tests.cpp:313-318 / common_data_utils.h:4349-4470: the 5x5 lower-stored complex matrix, xref = 1..5,
    b = op(T) xref built by the test itself, x = solve -> xref.  All six (fill, op) cases; with fill = upper the same
    arrays leave only the diagonal."""
    rp = np.array([0, 1, 3, 5, 7, 11], np.int32)
    ci = np.array([0, 0, 1, 1, 2, 2, 3, 0, 1, 2, 4], np.int32)
    v = np.array([4 - 0j, 2 + 2j, 5 - 1j, 1 + 2j, 3 - 3j, 2 + 0.5j, 4 - 4j, 1 + 2j, 3 - 3j, 0 - 2j, 2 - 2j])
    dense = np.zeros((5, 5), np.complex128)
    for i in range(5):
        dense[i, ci[rp[i]:rp[i + 1]]] = v[rp[i]:rp[i + 1]]
    xref = np.arange(1.0, 6.0) + 0j
    for base, fill, op in ((0, "lower", "n"), (0, "lower", "t"), (0, "lower", "h"), (1, "upper", "n"), (1, "upper", "t"), (1, "upper", "h")):
        rpb, cib = rp + base, ci + base
        h = ctypes.c_void_p()
        assert L.aoclsparse_create_zcsr(ctypes.byref(h), base, 5, 5, 11, P._ptr(rpb), P._ptr(cib), P._ptr(v)) == 0
        d = P.Descr(base=base, mtype=P.TYPE_TRIANGULAR, fill=P.FILL_LOWER if fill == "lower" else P.FILL_UPPER)
        b = _op_tri(dense, fill, "non_unit", op) @ xref
        x = np.zeros(5, np.complex128)
        o = {"n": P.OP_NONE, "t": P.OP_TRANSPOSE, "h": P.OP_CONJ_TRANSPOSE}[op]
        assert L.aoclsparse_ztrsv(o, P.CDouble(1, 0), h, d.h, P._ptr(b), P._ptr(x)) == 0
        assert np.max(np.abs(x - xref)) <= 10 * np.sqrt(2 * EPS64)  # the reference's own acceptance (expected_precision)
        assert np.max(np.abs(x - xref)) <= 32 * EPS64 * 5
        L.aoclsparse_destroy(ctypes.byref(h))


@pytest.mark.parametrize("prec", ["z", "c"])
def test_complex_csrmm(prec):
    """aoclsparse_{c,z}csrmm: general (N / T / H, rectangular), symmetric and hermitian, both layouts with padded
    leading dimensions, alpha == 0 and beta == 0 paths; against the dense product of the assembled operator."""
    dtype, eps = (np.complex128, EPS64) if prec == "z" else (np.complex64, EPS32)
    create = L.aoclsparse_create_zcsr if prec == "z" else L.aoclsparse_create_ccsr
    mm = L.aoclsparse_zcsrmm if prec == "z" else L.aoclsparse_ccsrmm
    CT = P.CDouble if prec == "z" else P.CFloat
    rng = np.random.default_rng(121)
    ops = {"n": P.OP_NONE, "t": P.OP_TRANSPOSE, "h": P.OP_CONJ_TRANSPOSE}
    for (m, k, types) in ((310, 270, ["general"]), (260, 260, ["symmetric", "hermitian"])):
        rp, ci, v = _cplx_matrix(122, m, k, 15, dtype)
        if m == k:
            dense = np.zeros((m, k), dtype)
            for i in range(m):
                dense[i, ci[rp[i]:rp[i + 1]]] = v[rp[i]:rp[i + 1]]
                dense[i, i] = 2.0
            rows = [np.flatnonzero(dense[i]) for i in range(m)]
            rp = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int32)
            ci = np.concatenate(rows).astype(np.int32)
            v = np.concatenate([dense[i, r] for i, r in enumerate(rows)]).astype(dtype)
        h = ctypes.c_void_p()
        assert create(ctypes.byref(h), 0, m, k, len(v), P._ptr(rp), P._ptr(ci), P._ptr(v)) == 0
        for mtype in types:
            d = P.Descr(mtype={"general": 0, "symmetric": 1, "hermitian": 2}[mtype], fill=P.FILL_UPPER)
            for opn, op in ops.items():
                # assembled operator through the SpMV restatement applied to unit vectors would be slow: build it once
                cols = k if (opn == "n" or mtype != "general") else m
                eye = np.eye(cols, dtype=np.complex128)
                Mo = np.stack([oracle.zmv(opn, mtype, "upper", "non_unit", 0, 1.0, m, k, rp, ci, v, eye[:, j], 0.0,
                                          np.zeros(m if (opn == "n" or mtype != "general") else k))[0] for j in range(cols)], axis=1)
                mc, kb, n = Mo.shape[0], Mo.shape[1], 37
                for order, colmaj in ((P.ORDER_ROW, False), (P.ORDER_COLUMN, True)):
                    ldb, ldc = (kb + 3, mc + 2) if colmaj else (n + 5, n + 1)
                    Bm = (rng.uniform(-1, 1, (kb, n)) + 1j * rng.uniform(-1, 1, (kb, n))).astype(dtype)
                    C0 = (rng.uniform(-1, 1, (mc, n)) + 1j * rng.uniform(-1, 1, (mc, n))).astype(dtype)
                    def pack(M, ld):
                        buf = np.full((M.shape[1], ld) if colmaj else (M.shape[0], ld), 7 + 7j, dtype)
                        if colmaj:
                            buf[:, :M.shape[0]] = M.T
                        else:
                            buf[:, :M.shape[1]] = M
                        return buf
                    def unpack(buf, r, c):
                        return buf[:, :r].T.copy() if colmaj else buf[:, :c].copy()
                    for alpha, beta in ((0.6 - 0.8j, -0.5 + 0.25j), (1.0 + 0j, 0j), (0j, 2.0 - 1j)):
                        Bb, Cb = pack(Bm, ldb), pack(C0, ldc)
                        if beta == 0:
                            # default: 0 * C is computed as in the reference, NaN stays; overwrite mode: never read
                            Cn = np.full_like(Cb, np.nan + 1j * np.nan)
                            assert mm(op, CT(alpha.real, alpha.imag), h, d.h, order, P._ptr(Bb), n, ldb, CT(0.0, 0.0),
                                      P._ptr(Cn), ldc) == 0
                            assert np.all(np.isnan(unpack(Cn, mc, n)))
                            Cb[...] = np.nan + 1j * np.nan
                            assert L.aoclsparse_mi355_set_csrmm_beta0_overwrite(1) == 0
                        try:
                            assert mm(op, CT(alpha.real, alpha.imag), h, d.h, order, P._ptr(Bb), n, ldb, CT(beta.real, beta.imag),
                                      P._ptr(Cb), ldc) == 0, (mtype, opn, colmaj)
                        finally:
                            assert L.aoclsparse_mi355_set_csrmm_beta0_overwrite(0) == 0
                        got = unpack(Cb, mc, n)
                        ref = alpha * (Mo @ Bm.astype(np.complex128)) + (beta * C0 if beta != 0 else 0)
                        scale = abs(alpha) * (np.abs(Mo) @ np.abs(Bm)) + abs(beta) * np.abs(C0)
                        assert np.all(np.abs(got - ref) <= 80 * eps * (scale + 1e-30)), (prec, mtype, opn, colmaj, alpha, beta)
                        pad = Cb[:, mc:] if colmaj else Cb[:, n:]
                        assert beta == 0 or np.all(pad == 7 + 7j)
        L.aoclsparse_destroy(ctypes.byref(h))


# --------------------------------------------------------------------------------------------------
# BASELINE configs 4 and 5 at full size, through size-independent properties
# --------------------------------------------------------------------------------------------------
def test_csrmm_full_config_properties():
    """Config 4: A = 5-pt Laplacian on 1000^2 (1M x 1M), B 1M x 256 fp64.  A * ones has a closed form (exact in
    fp64), the product is linear in B (scaling by 2 is exact), and four columns are checked against the oracle."""
    g, n = 1000, 256
    m, rp, ci, v = laplace5(g)
    A = P.Matrix(0, m, m, rp, ci, v)
    d = P.Descr()
    assert L.aoclsparse_set_mm_hint(A.h, P.OP_NONE, d.h, 10) == 0 and L.aoclsparse_optimize(A.h) == 0
    deg = (np.diff(rp) - 1).astype(np.float64)
    for order, ld in ((P.ORDER_ROW, n), (P.ORDER_COLUMN, m)):
        B = torch.ones(m * n, dtype=torch.float64, device="cuda")
        # (finite garbage, not NaN: an exactly-zero result still reads C to get the reference's sign of zero)
        C = torch.full((m * n,), 7.0, dtype=torch.float64, device="cuda")
        assert P.dcsrmm(P.OP_NONE, 1.0, A, d, order, B, n, ld, 0.0, C, ld) == 0
        torch.cuda.synchronize()
        Cm = C.view(m, n) if order == P.ORDER_ROW else C.view(n, m).t()
        expect = torch.from_numpy(4.0 - deg).cuda()
        assert torch.equal(Cm[:, 0], expect) and torch.equal(Cm[:, n - 1], expect) and torch.equal(Cm.sum(dim=1), expect * n)
        del B, C, Cm
    gen = torch.Generator(device="cuda")
    gen.manual_seed(5)
    B = torch.rand(m * n, dtype=torch.float64, device="cuda", generator=gen)
    C1, C2 = torch.zeros_like(B), torch.zeros_like(B)
    assert P.dcsrmm(P.OP_NONE, 1.0, A, d, P.ORDER_COLUMN, B, n, m, 0.0, C1, m) == 0
    assert P.dcsrmm(P.OP_NONE, 1.0, A, d, P.ORDER_COLUMN, 2.0 * B, n, m, 0.0, C2, m) == 0
    torch.cuda.synchronize()
    assert torch.equal(C2, 2.0 * C1)
    for j in (0, 1, 100, 255):
        so, cr = oracle.dcsrmm("col", 1.0, 0, v, ci, rp, m, B[j * m:(j + 1) * m].cpu().numpy(), 1, m, 0.0, np.zeros(m), m)
        assert np.array_equal(C1[j * m:(j + 1) * m].cpu().numpy(), cr)


def test_trsv_full_config_properties():
    """Config 5 in its Laplacian form: unit-lower ILU(0) factor of the 1000^2 grid (1M rows, 1,999 levels).  The
    solve is bit-identical to the serial CPU solve and L x reproduces b to a few ulps of the row scale."""
    g = 1000
    m, rp, ci, v = laplace5(g)
    st, lu, dg = oracle.dilu0(m, 0, rp, ci, v)
    assert st == 0
    A = P.Matrix(0, m, m, rp, ci, lu)
    dl = P.Descr(mtype=P.TYPE_TRIANGULAR, fill=P.FILL_LOWER, diag=P.DIAG_UNIT)
    assert L.aoclsparse_set_sv_hint(A.h, P.OP_NONE, dl.h, 10) == 0 and L.aoclsparse_optimize(A.h) == 0
    assert A.trsv_levels(P.FILL_LOWER) == 2 * g - 1
    b = np.random.default_rng(6).uniform(-1, 1, m)
    bd, xd = dev(b), torch.zeros(m, dtype=torch.float64, device="cuda")
    assert P.dtrsv(P.OP_NONE, 1.0, A, dl, bd, xd) == 0
    torch.cuda.synchronize()
    x = xd.cpu().numpy()
    assert np.array_equal(x, oracle_trsv(0, m, rp, ci, lu, "lower", "n", True, 1.0, b))
    # residual through the triangular product of the same handle: (L + I) x = b
    yd = torch.zeros(m, dtype=torch.float64, device="cuda")
    assert P.dmv(P.OP_NONE, 1.0, A, dl, xd, 0.0, yd) == 0
    torch.cuda.synchronize()
    r = np.abs(yd.cpu().numpy() - b)
    assert r.max() <= 16 * EPS64 * max(1.0, np.abs(x).max())


# --------------------------------------------------------------------------------------------------
# sparse x sparse with a dense result, CSR -> dense, sparse sum (sp2md.hpp, spmmd.cpp, convert.hpp:658-929, csradd.hpp)
# --------------------------------------------------------------------------------------------------
def test_spmmd_and_csr2dense_reference_kats(kats):
    """spmmd_tests.cpp:130-157 and conversion_tests.cpp:211-252 through the C ABI."""
    k = kats["spmmd"]
    for ba in (0, 1):
        for bb in (0, 1):
            A = P.Matrix(ba, 3, 3, np.array(k["a"]["row_ptr"], np.int32) + ba, np.array(k["a"]["col_ind"], np.int32) + ba,
                         np.array(k["a"]["val"], np.float64))
            B = P.Matrix(bb, 3, 3, np.array(k["b"]["row_ptr"], np.int32) + bb, np.array(k["b"]["col_ind"], np.int32) + bb,
                         np.array(k["b"]["val"], np.float64))
            for op, gold in ((P.OP_NONE, k["c_none"]), (P.OP_TRANSPOSE, k["c_trans"]), (P.OP_CONJ_TRANSPOSE, k["c_trans"])):
                G = np.array(gold, np.float64).reshape(3, 3)
                C = np.full(9, np.nan)
                assert L.aoclsparse_dspmmd(op, A.h, B.h, P.ORDER_ROW, P._ptr(C), 3) == 0
                assert np.array_equal(C.reshape(3, 3), G)
                C = np.full(15, 7.0)
                assert L.aoclsparse_dspmmd(op, A.h, B.h, P.ORDER_COLUMN, P._ptr(C), 5) == 0
                assert np.array_equal(C.reshape(3, 5)[:, :3].T, G) and np.all(C.reshape(3, 5)[:, 3:] == 7.0)
    k = kats["csr2dense"]
    rp, ci, v = np.array(k["row_ptr"], np.int32), np.array(k["col_ind"], np.int32), np.array(k["val"], np.float64)
    d = P.Descr()
    for gold, ld, order in ((k["rowmajor"], 5, P.ORDER_ROW), (k["colmajor"], 5, P.ORDER_COLUMN), (k["rowmajor_ld8"], 8, P.ORDER_ROW)):
        D = np.zeros(5 * ld) if ld == 8 else np.full(25, -1.0)
        assert L.aoclsparse_dcsr2dense(5, 5, d.h, P._ptr(v), P._ptr(rp), P._ptr(ci), P._ptr(D), ld, order) == 0
        assert np.array_equal(D, np.array(gold, np.float64))
    D, d1 = np.full(25, -1.0), P.Descr(base=1)
    rp1, ci1 = rp + 1, ci + 1
    assert L.aoclsparse_dcsr2dense(5, 5, d1.h, P._ptr(v), P._ptr(rp1), P._ptr(ci1), P._ptr(D), 5, P.ORDER_ROW) == 0
    assert np.array_equal(D, np.array(k["rowmajor"], np.float64))
    Df = np.full(25, -1.0, np.float32)
    vf = v.astype(np.float32)
    assert L.aoclsparse_scsr2dense(5, 5, d.h, P._ptr(vf), P._ptr(rp), P._ptr(ci), P._ptr(Df), 5, P.ORDER_COLUMN) == 0
    assert np.array_equal(Df, np.array(k["colmajor"], np.float32))


@pytest.mark.parametrize("base_a,base_b", [(0, 0), (1, 0), (0, 1)])
def test_sp2md_bit_exact_vs_oracle(base_a, base_b):
    """Every (opA, opB) in {N, T}^2, both layouts, padded ldc, alpha / beta incl. 0 and 1: the dense result carries the
    reference's per-element chain (unsorted operand rows; one long row of A)."""
    m, k, n = 700, 520, 610
    pa, ia, va = random_csr(401, m, k, lambda r, i: 300 if i == 5 else r.integers(0, 9), base=base_a, sort=False)
    pb, ib, vb = random_csr(402, k, n, lambda r, i: r.integers(0, 80) if i % 50 == 0 else r.integers(0, 9), base=base_b, sort=False)
    pbt, ibt, vbt = random_csr(403, n, k, lambda r, i: r.integers(0, 9), base=base_b, sort=False)  # for op(B) = T
    pa2, ia2, va2 = random_csr(404, k, m, lambda r, i: r.integers(0, 9), base=base_a, sort=False)  # for op(A) = T
    A, B = P.Matrix(base_a, m, k, pa, ia, va), P.Matrix(base_b, k, n, pb, ib, vb)
    Bt, A2 = P.Matrix(base_b, n, k, pbt, ibt, vbt), P.Matrix(base_a, k, m, pa2, ia2, va2)
    dA, dB = P.Descr(base=base_a), P.Descr(base=base_b)
    a, b, bt, a2 = (m, k, base_a, pa, ia, va), (k, n, base_b, pb, ib, vb), (n, k, base_b, pbt, ibt, vbt), (k, m, base_a, pa2, ia2, va2)
    rng = np.random.default_rng(9)
    cases = [(A, a, False, B, b, False), (A2, a2, True, B, b, False), (A, a, False, Bt, bt, True), (A2, a2, True, Bt, bt, True)]
    for (HA, ta, trA, HB, tb, trB) in cases:
        for rowmaj, pad in ((True, 0), (False, 0), (True, 3), (False, 5)):
            for alpha, beta in ((1.0, 0.0), (-1.25, 0.5), (2.0, 1.0), (0.0, 3.0)):
                ldc = (n if rowmaj else m) + pad
                outer = m if rowmaj else n
                C0 = rng.uniform(-1, 1, outer * ldc)
                if beta == 0.0:
                    C0.reshape(outer, ldc)[:, :ldc - pad] = np.nan  # never read
                want = oracle.dsp2md(ta, trA, tb, trB, alpha, beta, C0, rowmaj, ldc)
                C = C0.copy()
                st = L.aoclsparse_dsp2md(P.OP_TRANSPOSE if trA else P.OP_NONE, dA.h, HA.h, P.OP_TRANSPOSE if trB else P.OP_NONE,
                                         dB.h, HB.h, alpha, beta, P._ptr(C), P.ORDER_ROW if rowmaj else P.ORDER_COLUMN, ldc)
                assert st == 0 and np.array_equal(C, want), (trA, trB, rowmaj, pad, alpha, beta)
    # device-resident result
    import torch
    Cd = torch.zeros(m * n, dtype=torch.float64, device="cuda")
    assert L.aoclsparse_dspmmd(P.OP_NONE, A.h, B.h, P.ORDER_ROW, ctypes.c_void_p(Cd.data_ptr()), n) == 0
    assert np.array_equal(Cd.cpu().numpy(), oracle.dsp2md(a, False, b, False, 1.0, 0.0, np.zeros(m * n), True, n))


@pytest.mark.parametrize("prec", ["s", "c", "z"])
def test_sp2md_float_and_complex(prec):
    """fp32 and complex sp2md / spmmd: within (terms + 4) eps sum|a||b| of the dense product for every op pair (H
    conjugates as it transposes, sp2md.hpp:281-347); tolerance: 16 eps * (|alpha| |A||B| + |beta C|)."""
    m, k, n = 150, 130, 140
    rng = np.random.default_rng(31)
    cplx = prec != "s"
    dtype = {"s": np.float32, "c": np.complex64, "z": np.complex128}[prec]
    eps = EPS64 if prec == "z" else EPS32
    create = getattr(L, f"aoclsparse_create_{prec}csr")
    S = {"s": ctypes.c_float, "c": P.CFloat, "z": P.CDouble}[prec]
    sc = (lambda z: S(z.real, z.imag)) if cplx else (lambda z: S(z))

    def mat(seed, r, c):
        rp, ci, vr = random_csr(seed, r, c, lambda g, i: g.integers(0, 9), sort=False)
        v = (vr + (1j * rng.uniform(-1, 1, len(vr)) if cplx else 0)).astype(dtype)
        h = ctypes.c_void_p()
        assert create(ctypes.byref(h), 0, r, c, len(v), P._ptr(rp), P._ptr(ci), P._ptr(v)) == 0
        D = np.zeros((r, c), np.complex128 if cplx else np.float64)
        for i in range(r):
            D[i, ci[rp[i]:rp[i + 1]]] = v[rp[i]:rp[i + 1]]
        return h, D, (rp, ci, v)

    f = {P.OP_NONE: lambda X: X, P.OP_TRANSPOSE: lambda X: X.T, P.OP_CONJ_TRANSPOSE: lambda X: X.conj().T}
    d = P.Descr()
    keep = []
    alpha, beta = (0.75 - 0.5j, -0.25 + 1.0j) if cplx else (0.75, -0.25)
    for opA in f:
        for opB in f:
            hA, DA, ka = mat(1, *((m, k) if opA == P.OP_NONE else (k, m)))
            hB, DB, kb = mat(2, *((k, n) if opB == P.OP_NONE else (n, k)))
            keep += [ka, kb]
            for rowmaj in (True, False):
                C0 = (rng.uniform(-1, 1, (m, n)) + (1j * rng.uniform(-1, 1, (m, n)) if cplx else 0)).astype(dtype)
                C = np.ascontiguousarray(C0 if rowmaj else C0.T).copy()
                st = getattr(L, f"aoclsparse_{prec}sp2md")(opA, d.h, hA, opB, d.h, hB, sc(alpha), sc(beta), P._ptr(C),
                                                           P.ORDER_ROW if rowmaj else P.ORDER_COLUMN, n if rowmaj else m)
                got = C if rowmaj else C.T
                want = alpha * (f[opA](DA) @ f[opB](DB)) + beta * C0
                bound = 16 * eps * (abs(alpha) * (np.abs(f[opA](DA)) @ np.abs(f[opB](DB))) + abs(beta) * np.abs(C0)) + 1e-30
                assert st == 0 and np.all(np.abs(got - want) <= bound), (opA, opB, rowmaj)
            if opB == P.OP_NONE:
                C = np.full((m, n), np.nan, dtype)
                assert getattr(L, f"aoclsparse_{prec}spmmd")(opA, hA, hB, P.ORDER_ROW, P._ptr(C), n) == 0
                assert np.all(np.abs(C - f[opA](DA) @ DB) <= 16 * eps * (np.abs(f[opA](DA)) @ np.abs(DB)) + 1e-30)
            L.aoclsparse_destroy(ctypes.byref(hA))
            L.aoclsparse_destroy(ctypes.byref(hB))


def test_csr2dense_all_descriptors_vs_oracle():
    """General / symmetric / hermitian / triangular, both fills, the three diagonal kinds, both layouts where the
    reference offers them, padded ld, device-resident arrays; complex hermitian mirrors conjugate."""
    import torch
    m = 333
    for base in (0, 1):
        rp, ci, v = random_csr(501, m, m, lambda r, i: r.integers(0, 12), base=base, sort=False)
        for mode, mtype in ((0, P.TYPE_GENERAL), (1, P.TYPE_SYMMETRIC), (3, P.TYPE_TRIANGULAR)):
            for fill in (0, 1):
                for diag in (0, 1, 2):
                    for colmaj in (False, True):
                        if colmaj and mode == 3:
                            continue
                        d = P.Descr(base=base, mtype=mtype, fill=fill, diag=diag)
                        ld = m + 2
                        D0 = np.random.default_rng(3).uniform(-1, 1, m * ld)
                        want = oracle.dcsr2dense(m, m, base, rp, ci, v, D0, ld, colmaj, mode, fill, diag)
                        D = D0.copy()
                        st = L.aoclsparse_dcsr2dense(m, m, d.h, P._ptr(v), P._ptr(rp), P._ptr(ci), P._ptr(D), ld,
                                                     P.ORDER_COLUMN if colmaj else P.ORDER_ROW)
                        assert st == 0 and np.array_equal(D, want), (base, mode, fill, diag, colmaj)
    # rectangular general, device-resident CSR and dense arrays
    rp, ci, v = random_csr(502, 400, 250, lambda r, i: r.integers(0, 30), sort=False)
    t = [torch.from_numpy(x).cuda() for x in (v, rp, ci)]
    Dd = torch.full((400 * 250,), -1.0, dtype=torch.float64, device="cuda")
    assert L.aoclsparse_dcsr2dense(400, 250, P.Descr().h, *[ctypes.c_void_p(x.data_ptr()) for x in t],
                                   ctypes.c_void_p(Dd.data_ptr()), 250, P.ORDER_ROW) == 0
    assert np.array_equal(Dd.cpu().numpy().reshape(400, 250), _dense(400, 250, rp, ci, v))
    # complex hermitian from the lower triangle
    rp, ci, vr = random_csr(503, 90, 90, lambda r, i: r.integers(1, 9))
    vz = (vr + 1j * np.random.default_rng(4).uniform(-1, 1, len(vr))).astype(np.complex128)
    Dz = np.zeros((90, 90), np.complex128)
    d = P.Descr(mtype=P.TYPE_HERMITIAN, fill=0, diag=0)
    assert L.aoclsparse_zcsr2dense(90, 90, d.h, P._ptr(vz), P._ptr(rp), P._ptr(ci), P._ptr(Dz), 90, P.ORDER_ROW) == 0
    F = np.zeros((90, 90), np.complex128)
    for i in range(90):
        F[i, ci[rp[i]:rp[i + 1]]] = vz[rp[i]:rp[i + 1]]
    lo = np.tril(F, -1)
    assert np.array_equal(Dz, lo + lo.conj().T + np.diag(np.diag(F)))


@pytest.mark.parametrize("base_a,base_b", [(0, 0), (1, 0), (0, 1), (1, 1)])
def test_add_bit_exact_vs_oracle(base_a, base_b):
    """C = alpha*op(A) + B: row_ptr / col_ind (A's row, then B's new columns in B's order, A's base) and values are the
    reference's, bit for bit; op = N and T; rows longer than a wavefront; alpha = 0 keeps A's pattern."""
    m, n = 2500, 2100
    pa, ia, va = random_csr(601, m, n, lambda r, i: 200 if i == 3 else r.integers(0, 10), base=base_a, sort=False)
    pb, ib, vb = random_csr(602, m, n, lambda r, i: 150 if i in (3, 4) else r.integers(0, 10), base=base_b, sort=False)
    pt, it, vt = random_csr(603, n, m, lambda r, i: r.integers(0, 10), base=base_a, sort=False)
    A, B, At = P.Matrix(base_a, m, n, pa, ia, va), P.Matrix(base_b, m, n, pb, ib, vb), P.Matrix(base_a, n, m, pt, it, vt)
    for H, ta, op, alpha in ((A, (m, n, base_a, pa, ia, va), P.OP_NONE, -1.75), (At, (n, m, base_a, pt, it, vt), P.OP_TRANSPOSE, 0.3),
                             (A, (m, n, base_a, pa, ia, va), P.OP_NONE, 0.0)):
        pc, ic, vc = oracle.dcsradd(ta, op != P.OP_NONE, alpha, (m, n, base_b, pb, ib, vb))
        C = ctypes.c_void_p()
        assert L.aoclsparse_dadd(op, H.h, alpha, B.h, ctypes.byref(C)) == 0
        b, cm, cn, cz, row, col, val = _export(C)
        assert (b, cm, cn, cz) == (base_a, m, n, len(ic))
        assert np.array_equal(row, pc) and np.array_equal(col, ic) and np.array_equal(val, vc)
        # the sum is an ordinary handle: it multiplies
        x = np.random.default_rng(8).uniform(-1, 1, n)
        y = np.zeros(m)
        dC = P.Descr(base=base_a)
        one, zero = ctypes.c_double(1.0), ctypes.c_double(0.0)
        assert L.aoclsparse_mi355_set_option(P.OPTION_SPMV_STRICT, 1) == 0  # rows of 200-350 entries: the reference's chain
        try:
            assert L.aoclsparse_dmv(P.OP_NONE, ctypes.byref(one), C, dC.h, P._ptr(x), ctypes.byref(zero), P._ptr(y)) == 0
        finally:
            assert L.aoclsparse_mi355_set_option(P.OPTION_SPMV_STRICT, 0) == 0
        so, yr = oracle.dcsrmv(-1, base_a, 1.0, m, len(vc), vc, ic, pc, x, 0.0, np.zeros(m))
        assert np.array_equal(y, yr)
        assert L.aoclsparse_destroy(ctypes.byref(C)) == 0
    # fp32 and complex (conjugate transpose) against the dense sum
    Af, Bf = P.Matrix(base_a, m, n, pa, ia, va.astype(np.float32)), P.Matrix(base_b, m, n, pb, ib, vb.astype(np.float32))
    C = ctypes.c_void_p()
    assert L.aoclsparse_sadd(P.OP_NONE, Af.h, 0.5, Bf.h, ctypes.byref(C)) == 0
    _, _, _, cz, row, col, val = _export(C, double=False)
    pc, ic, vc = oracle.dcsradd((m, n, base_a, pa, ia, va), False, 0.5, (m, n, base_b, pb, ib, vb))
    assert np.array_equal(row, pc) and np.array_equal(col, ic)
    assert np.allclose(val, vc, rtol=4 * EPS32, atol=4 * EPS32)
    L.aoclsparse_destroy(ctypes.byref(C))
    vz, wz = (vt + 1j * vt[::-1]).astype(np.complex128), (vb - 0.5j * vb).astype(np.complex128)
    hz, hw = ctypes.c_void_p(), ctypes.c_void_p()
    assert L.aoclsparse_create_zcsr(ctypes.byref(hz), base_a, n, m, len(vz), P._ptr(pt), P._ptr(it), P._ptr(vz)) == 0
    assert L.aoclsparse_create_zcsr(ctypes.byref(hw), base_b, m, n, len(wz), P._ptr(pb), P._ptr(ib), P._ptr(wz)) == 0
    assert L.aoclsparse_zadd(P.OP_CONJ_TRANSPOSE, hz, P.CDouble(0.5, -2.0), hw, ctypes.byref(C)) == 0
    bz, mz, nz, zz = ctypes.c_int(), ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
    a1, a2, a3 = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
    assert L.aoclsparse_export_zcsr(C, ctypes.byref(bz), ctypes.byref(mz), ctypes.byref(nz), ctypes.byref(zz), ctypes.byref(a1),
                                    ctypes.byref(a2), ctypes.byref(a3)) == 0
    row = np.ctypeslib.as_array(ctypes.cast(a1, ctypes.POINTER(ctypes.c_int32)), (m + 1,)) - base_a
    col = np.ctypeslib.as_array(ctypes.cast(a2, ctypes.POINTER(ctypes.c_int32)), (zz.value,)) - base_a
    val = np.ctypeslib.as_array(ctypes.cast(a3, ctypes.POINTER(ctypes.c_double)), (2 * zz.value,)).view(np.complex128)
    got = np.zeros((m, n), np.complex128)
    for i in range(m):
        got[i, col[row[i]:row[i + 1]]] = val[row[i]:row[i + 1]]
    Z = np.zeros((n, m), np.complex128)
    for i in range(n):
        Z[i, it[pt[i] - base_a:pt[i + 1] - base_a] - base_a] = vz[pt[i] - base_a:pt[i + 1] - base_a]
    Wd = np.zeros((m, n), np.complex128)
    for i in range(m):
        Wd[i, ib[pb[i] - base_b:pb[i + 1] - base_b] - base_b] = wz[pb[i] - base_b:pb[i + 1] - base_b]
    assert (mz.value, nz.value, bz.value) == (m, n, base_a)
    assert np.allclose(got, (0.5 - 2.0j) * Z.conj().T + Wd, rtol=0, atol=8 * EPS64)
    for h in (C, hz, hw):
        L.aoclsparse_destroy(ctypes.byref(h))


def test_spmmd_full_size_property_on_the_device():
    """L64^2 (4096 x 4096, dense result 134 MB, device-resident): (A*A) 1 equals A (A 1) -- integer-valued, so exactly --
    and the result is symmetric; a second call reuses the handles' device copies and gives the same bits."""
    import torch
    m, rp, ci, v = laplace5(64)
    A = P.Matrix(0, m, m, rp, ci, v)
    Cd = torch.full((m * m,), float("nan"), dtype=torch.float64, device="cuda")
    assert L.aoclsparse_dspmmd(P.OP_NONE, A.h, A.h, P.ORDER_COLUMN, ctypes.c_void_p(Cd.data_ptr()), m) == 0
    C = Cd.view(m, m)
    one = np.ones(m)
    so, y1 = oracle.dcsrmv(-1, 0, 1.0, m, len(v), v, ci, rp, one, 0.0, np.zeros(m))
    so, y2 = oracle.dcsrmv(-1, 0, 1.0, m, len(v), v, ci, rp, y1, 0.0, np.zeros(m))
    assert np.array_equal(C.sum(dim=0).cpu().numpy(), y2) and bool(torch.equal(C, C.t()))
    C2 = torch.zeros_like(Cd)
    assert L.aoclsparse_dspmmd(P.OP_TRANSPOSE, A.h, A.h, P.ORDER_ROW, ctypes.c_void_p(C2.data_ptr()), m) == 0
    assert bool(torch.equal(C2, Cd))


# --------------------------------------------------------------------------------------------------
# level 1: compressed sparse vector against a dense vector (level1/aoclsparse_{axpyi,dot,gthr,sctr,roti}.hpp)
# --------------------------------------------------------------------------------------------------
def test_level1_reference_kats(kats):
    """axpyi_tests.cpp:78-88, roti_tests.cpp:50-99, dotp_tests.cpp:64-69, gthr_tests.cpp:66-71, sctr_tests.cpp:71-84
    through the C ABI with host vectors, double and float."""
    k = kats["level1"]
    for dt, p, eps in ((np.float64, "d", EPS64), (np.float32, "s", EPS32)):
        a = k["axpyi"]
        for nnz, gold in ((4, a["y_nnz4"]), (2, a["y_nnz2"])):
            x, ix, y = np.array(a["x"], dt), np.array(a["indx"], np.int32), np.array(a["y"], dt)
            assert getattr(L, f"aoclsparse_{p}axpyi")(nnz, a["a"], P._ptr(x), P._ptr(ix), P._ptr(y)) == 0
            assert np.array_equal(y, np.array(gold, dt))
        for r in k["roti"]:
            x, ix, y = np.array(r["x"], dt), np.array(r["indx"], np.int32), np.array(r["y"], dt)
            assert getattr(L, f"aoclsparse_{p}roti")(len(ix), P._ptr(x), P._ptr(ix), P._ptr(y), r["c"], r["s"]) == 0
            assert np.array_equal(x, np.array(r["x_exp"], dt)) and np.array_equal(y, np.array(r["y_exp"], dt))
        d = k["doti"]
        x, ix, y = np.array(d["x"], dt), np.array(d["indx"], np.int32), np.array(d["y"], dt)
        got = getattr(L, f"aoclsparse_{p}doti")(len(ix), P._ptr(x), P._ptr(ix), P._ptr(y))
        assert abs(got - d["dot"]) <= 64 * eps * np.sum(np.abs(x * y[ix]))
        g = k["gthr"]
        ix = np.array(g["indx"], np.int32)
        for fn, ygold in (("gthr", g["y"]), ("gthrz", g["y_gthrz"])):
            y, x = np.array(g["y"], dt), np.full(len(ix), -1, dt)
            assert getattr(L, f"aoclsparse_{p}{fn}")(len(ix), P._ptr(y), P._ptr(x), P._ptr(ix)) == 0
            assert np.array_equal(x, np.array(g["x_exp"], dt)) and np.array_equal(y, np.array(ygold, dt))
        s = k["sctr"]
        for nnz, gold in ((17, s["y_nnz17"]), (10, s["y_nnz10"])):
            x, ix, y = np.array(s["x"], dt), np.array(s["indx"], np.int32), np.zeros(17, dt)
            assert getattr(L, f"aoclsparse_{p}sctr")(nnz, P._ptr(x), P._ptr(ix), P._ptr(y)) == 0
            assert np.array_equal(y, np.array(gold, dt))
    y = np.arange(12, dtype=np.float64)
    assert L.aoclsparse_daxpyi(3, 1.0, P._ptr(np.ones(3)), P._ptr(np.array([1, -2, 3], np.int32)), P._ptr(y)) == 6
    assert np.array_equal(y, np.arange(12))  # nothing is modified when an index is negative


def test_level1_large_host_and_device_vectors():
    """2 M entries into an 8 M vector: axpyi bit-exact against the oracle (one contracted multiply-add per entry),
    gather / scatter / strided forms exact, roti against the oracle's contraction, dot products within
    n eps sum|x y| of the serial sum; host vectors and device vectors; complex twins against numpy."""
    import torch
    n, nnz = 1 << 23, 1 << 21
    rng = np.random.default_rng(71)
    ix = rng.permutation(n)[:nnz].astype(np.int32)
    x, y = rng.uniform(-1, 1, nnz), rng.uniform(-1, 1, n)
    st, yr = oracle.daxpyi(0.3, x, ix, y)
    y1 = y.copy()
    assert L.aoclsparse_daxpyi(nnz, 0.3, P._ptr(x), P._ptr(ix), P._ptr(y1)) == 0 and np.array_equal(y1, yr)
    xd, ixd, yd = dev(x), torch.from_numpy(ix).cuda(), dev(y)
    dp = lambda t: ctypes.c_void_p(t.data_ptr())
    assert L.aoclsparse_daxpyi_kid(nnz, 0.3, dp(xd), dp(ixd), dp(yd), 3) == 0
    torch.cuda.synchronize()
    assert np.array_equal(yd.cpu().numpy(), yr)
    # dot: device vectors, result by value; host vectors
    ref = oracle.ddoti(x, ix, y)
    bound = nnz * EPS64 * np.sum(np.abs(x * y[ix]))
    assert abs(L.aoclsparse_ddoti(nnz, dp(xd), dp(ixd), dp(dev(y))) - ref) <= bound
    assert abs(L.aoclsparse_ddoti_kid(nnz, P._ptr(x), P._ptr(ix), P._ptr(y), 0) - ref) <= bound
    assert L.aoclsparse_ddoti(nnz, dp(xd), dp(ixd), dp(dev(y))) == L.aoclsparse_ddoti(nnz, P._ptr(x), P._ptr(ix), P._ptr(y))
    # gather, gather-and-zero, strided gather / scatter, scatter
    xg = np.zeros(nnz)
    assert L.aoclsparse_dgthr(nnz, P._ptr(y), P._ptr(xg), P._ptr(ix)) == 0 and np.array_equal(xg, y[ix])
    y2, xg = y.copy(), np.zeros(nnz)
    assert L.aoclsparse_dgthrz(nnz, P._ptr(y2), P._ptr(xg), P._ptr(ix)) == 0
    yz = y.copy()
    yz[ix] = 0
    assert np.array_equal(xg, y[ix]) and np.array_equal(y2, yz)
    xs = np.zeros(nnz)
    assert L.aoclsparse_dgthrs(nnz, P._ptr(y), P._ptr(xs), 3) == 0 and np.array_equal(xs, y[0:3 * nnz:3])
    y3 = y.copy()
    assert L.aoclsparse_dsctrs(nnz, P._ptr(x), 4, P._ptr(y3)) == 0
    yw = y.copy()
    yw[0:4 * nnz:4] = x
    assert np.array_equal(y3, yw)
    y4d = dev(y)
    assert L.aoclsparse_dsctr(nnz, dp(xd), dp(ixd), dp(y4d)) == 0
    torch.cuda.synchronize()
    assert np.array_equal(y4d.cpu().numpy(), oracle.sctr(x, ix, y))
    # Givens rotation
    st, xr, yr2 = oracle.droti(x, ix, y, 0.6, -0.8)
    x5, y5 = x.copy(), y.copy()
    assert L.aoclsparse_droti(nnz, P._ptr(x5), P._ptr(ix), P._ptr(y5), 0.6, -0.8) == 0
    assert np.array_equal(x5, xr) and np.array_equal(y5, yr2)
    # float and complex twins
    xf, yf = x.astype(np.float32), y.astype(np.float32)
    y6 = yf.copy()
    assert L.aoclsparse_saxpyi(nnz, 0.3, P._ptr(xf), P._ptr(ix), P._ptr(y6)) == 0
    want = yf.copy()
    want[ix] = (np.float32(0.3) * xf.astype(np.float64) + yf[ix].astype(np.float64)).astype(np.float32)  # exact product, one rounding
    assert np.array_equal(y6, want)
    m = 1 << 16
    iz = rng.permutation(1 << 18)[:m].astype(np.int32)
    xz = (rng.uniform(-1, 1, m) + 1j * rng.uniform(-1, 1, m)).astype(np.complex128)
    yz = (rng.uniform(-1, 1, 1 << 18) + 1j * rng.uniform(-1, 1, 1 << 18)).astype(np.complex128)
    az = np.array([0.5 - 1.5j])
    y7 = yz.copy()
    assert L.aoclsparse_zaxpyi(m, P._ptr(az), P._ptr(xz), P._ptr(iz), P._ptr(y7)) == 0
    w7 = yz.copy()
    w7[iz] += az[0] * xz
    assert np.allclose(y7, w7, rtol=0, atol=8 * EPS64)
    dot = np.zeros(1, np.complex128)
    assert L.aoclsparse_zdotci(m, P._ptr(xz), P._ptr(iz), P._ptr(yz), P._ptr(dot)) == 0
    assert abs(dot[0] - np.sum(np.conj(xz) * yz[iz])) <= m * EPS64 * np.sum(np.abs(xz * yz[iz]))
    assert L.aoclsparse_zdotui(m, P._ptr(xz), P._ptr(iz), P._ptr(yz), P._ptr(dot)) == 0
    assert abs(dot[0] - np.sum(xz * yz[iz])) <= m * EPS64 * np.sum(np.abs(xz * yz[iz]))
    xc, yc = xz.astype(np.complex64), yz.astype(np.complex64)
    dotc = np.zeros(1, np.complex64)
    assert L.aoclsparse_cdotci(m, P._ptr(xc), P._ptr(iz), P._ptr(yc), P._ptr(dotc)) == 0
    assert abs(dotc[0] - np.sum(np.conj(xz) * yz[iz])) <= 4 * m * EPS32 * np.sum(np.abs(xz * yz[iz]))
    xo = np.zeros(m, np.complex128)
    y8 = yz.copy()
    assert L.aoclsparse_zgthrz(m, P._ptr(y8), P._ptr(xo), P._ptr(iz)) == 0 and np.array_equal(xo, yz[iz]) and np.all(y8[iz] == 0)
    y9 = np.zeros(1 << 18, np.complex64)
    assert L.aoclsparse_csctr(m, P._ptr(xc), P._ptr(iz), P._ptr(y9)) == 0 and np.array_equal(y9[iz], xc)


# --------------------------------------------------------------------------------------------------
# DIA and BSR products (level2/aoclsparse_diamv.hpp:34-70, level2/aoclsparse_bsrmv_kr.hpp:30-154)
# --------------------------------------------------------------------------------------------------
def test_diamv_bsrmv_reference_kat(kats):
    """diamv_tests.cpp:137-197 and bsrmv_tests.cpp:40-101: CSR -> DIA / BSR through the library's converters, then the
    product; host arrays, beta = 0 with NaN in y."""
    k = kats["dia_bsr"]
    for base in (0, 1):
        rp, ci, v = np.array(k["row_ptr"], np.int32) + base, np.array(k["col_ind"], np.int32) + base, np.array(k["val"], np.float64)
        d = P.Descr(base=base)
        nd = ctypes.c_int32()
        assert L.aoclsparse_csr2dia_ndiag(5, 5, d.h, 7, P._ptr(rp), P._ptr(ci), ctypes.byref(nd)) == 0 and nd.value == 3
        off, dv = np.zeros(3, np.int32), np.zeros(15)
        assert L.aoclsparse_dcsr2dia(5, 5, d.h, P._ptr(rp), P._ptr(ci), P._ptr(v), 3, P._ptr(off), P._ptr(dv)) == 0
        a, b = ctypes.c_double(1.0), ctypes.c_double(0.0)
        x, y = np.array(k["x"][:5], np.float64), np.full(5, np.nan)
        assert L.aoclsparse_ddiamv(P.OP_NONE, ctypes.byref(a), 5, 5, 7, P._ptr(dv), P._ptr(off), 3, d.h, P._ptr(x), ctypes.byref(b), P._ptr(y)) == 0
        assert np.array_equal(y, k["y_gold"][:5])
        bp, nnzb = np.zeros(4, np.int32), ctypes.c_int32()
        assert L.aoclsparse_csr2bsr_nnz(5, 5, d.h, P._ptr(rp), P._ptr(ci), 2, P._ptr(bp), ctypes.byref(nnzb)) == 0 and nnzb.value == 4
        bi, bv = np.zeros(4, np.int32), np.zeros(16)
        assert L.aoclsparse_dcsr2bsr(5, 5, d.h, P.ORDER_COLUMN, P._ptr(v), P._ptr(rp), P._ptr(ci), 2, P._ptr(bv), P._ptr(bp), P._ptr(bi)) == 0
        x, y = np.array(k["x"], np.float64), np.full(6, np.nan)
        assert L.aoclsparse_dbsrmv(P.OP_NONE, ctypes.byref(a), 3, 3, 2, P._ptr(bv), P._ptr(bi), P._ptr(bp), d.h, P._ptr(x), ctypes.byref(b), P._ptr(y)) == 0
        assert np.array_equal(y, k["y_gold"])


@pytest.mark.parametrize("base", [0, 1])
def test_diamv_bsrmv_bit_exact_vs_oracle(base):
    """Banded and random matrices, rectangular, (alpha, beta) incl. 1 / 0, every block size the reference has a kernel
    for (2..8, 16) and two it serves by the general loop (1, 11); host and device arrays; fp32 against fp64 with an
    n eps bound.  One lane per scalar row keeps the reference kernel's chain: bit-identical."""
    import torch
    dp = lambda t: ctypes.c_void_p(t.data_ptr())
    for (m, n, rl, seed) in ((5000, 4300, lambda r, i: r.integers(0, 12), 1), (3000, 3000, None, 2)):
        rp, ci, v = (random_csr(seed, m, n, rl, base=base, sort=False) if seed == 1 else _banded(m, 7, base))
        d = P.Descr(base=base)
        rng = np.random.default_rng(seed)
        nd, off, dv = oracle.csr2dia(m, n, base, rp, ci, v)
        if nd * m < 40_000_000:
            x, y0 = rng.uniform(-1, 1, n), rng.uniform(-1, 1, m)
            for alpha, beta in ((1.0, 0.0), (-0.75, 1.0), (2.5, -0.5)):
                a, b = ctypes.c_double(alpha), ctypes.c_double(beta)
                want = oracle.ddiamv(alpha, m, n, dv, off, x, beta, y0)
                y = y0.copy() if beta != 0 else np.full(m, np.nan)
                assert L.aoclsparse_ddiamv(P.OP_NONE, ctypes.byref(a), m, n, len(v), P._ptr(dv), P._ptr(off), nd, d.h, P._ptr(x), ctypes.byref(b), P._ptr(y)) == 0
                assert np.array_equal(y, want), (m, alpha, beta)
                t = [torch.from_numpy(z).cuda() for z in (dv, off, x, y0.copy())]
                assert L.aoclsparse_ddiamv_kid(P.OP_NONE, ctypes.byref(a), m, n, len(v), dp(t[0]), dp(t[1]), nd, d.h, dp(t[2]), ctypes.byref(b), dp(t[3]), 0, 0) == 0
                torch.cuda.synchronize()
                assert np.array_equal(t[3].cpu().numpy(), want)
            af, bf = ctypes.c_float(2.5), ctypes.c_float(-0.5)
            yf = y0.astype(np.float32)
            assert L.aoclsparse_sdiamv(P.OP_NONE, ctypes.byref(af), m, n, len(v), P._ptr(dv.astype(np.float32)), P._ptr(off), nd, d.h,
                                       P._ptr(x.astype(np.float32)), ctypes.byref(bf), P._ptr(yf)) == 0
            scale = 2.5 * np.abs(oracle.ddiamv(1.0, m, n, np.abs(dv), off, np.abs(x), 0.0, y0)) + np.abs(y0)
            assert np.all(np.abs(yf - oracle.ddiamv(2.5, m, n, dv, off, x, -0.5, y0)) <= (nd + 4) * EPS32 * scale + 1e-30)
        for dim in (1, 2, 3, 4, 5, 6, 7, 8, 11, 16):
            mb, nb = (m + dim - 1) // dim, (n + dim - 1) // dim
            bp, bi, bv = oracle.csr2bsr(m, n, base, rp, ci, v, dim, False)
            x, y0 = rng.uniform(-1, 1, nb * dim), rng.uniform(-1, 1, mb * dim)
            for alpha, beta in ((1.0, 0.0), (-0.75, 1.0), (2.5, -0.5)):
                a, b = ctypes.c_double(alpha), ctypes.c_double(beta)
                want = oracle.dbsrmv(alpha, mb, dim, base, bv, bi, bp, x, beta, y0)
                y = y0.copy() if beta != 0 else np.full(mb * dim, np.nan)
                assert L.aoclsparse_dbsrmv(P.OP_NONE, ctypes.byref(a), mb, nb, dim, P._ptr(bv), P._ptr(bi), P._ptr(bp), d.h, P._ptr(x), ctypes.byref(b), P._ptr(y)) == 0
                assert np.array_equal(y, want), (m, dim, alpha, beta)
            t = [torch.from_numpy(z).cuda() for z in (bv, bi, bp, x, y0.copy())]
            assert L.aoclsparse_dbsrmv(P.OP_NONE, ctypes.byref(a), mb, nb, dim, dp(t[0]), dp(t[1]), dp(t[2]), d.h, dp(t[3]), ctypes.byref(b), dp(t[4])) == 0
            torch.cuda.synchronize()
            assert np.array_equal(t[4].cpu().numpy(), want)
        bp, bi, bv = oracle.csr2bsr(m, n, base, rp, ci, v, 4, False)
        mb, nb = (m + 3) // 4, (n + 3) // 4
        x, y0 = rng.uniform(-1, 1, nb * 4), rng.uniform(-1, 1, mb * 4)
        af, bf = ctypes.c_float(2.5), ctypes.c_float(-0.5)
        yf = y0.astype(np.float32)
        assert L.aoclsparse_sbsrmv(P.OP_NONE, ctypes.byref(af), mb, nb, 4, P._ptr(bv.astype(np.float32)), P._ptr(bi), P._ptr(bp), d.h,
                                   P._ptr(x.astype(np.float32)), ctypes.byref(bf), P._ptr(yf)) == 0
        scale = 2.5 * oracle.dbsrmv(1.0, mb, 4, base, np.abs(bv), bi, bp, np.abs(x), 0.0, y0) + np.abs(y0)
        assert np.all(np.abs(yf - oracle.dbsrmv(2.5, mb, 4, base, bv, bi, bp, x, -0.5, y0)) <= 64 * EPS32 * scale + 1e-30)


def _banded(m, half, base):
    """m x m band of half-width `half` (2*half+1 diagonals), rows sorted."""
    rows = [np.arange(max(0, i - half), min(m, i + half + 1)) for i in range(m)]
    rp = np.zeros(m + 1, np.int64)
    rp[1:] = np.cumsum([len(r) for r in rows])
    ci = np.concatenate(rows)
    v = np.random.default_rng(12).uniform(-1, 1, len(ci))
    return (rp + base).astype(np.int32), (ci + base).astype(np.int32), v


# --------------------------------------------------------------------------------------------------
# forward SOR sweep (solvers/aoclsparse_sorv.hpp)
# --------------------------------------------------------------------------------------------------
def test_sorv_kats_and_bit_exact_sweeps(kats):
    """The reference's vectors (sorv_tests.cpp:366-414, sample_dsorv.cpp) within its tolerance, and level-scheduled
    sweeps on large matrices -- structurally unsymmetric, rows unsorted, both bases, host and device vectors -- bit-identical
    to the serial restatement; repeated sweeps converge on a diagonally dominant system."""
    import torch
    tol = 10 * np.sqrt(2 * EPS64)
    for k in kats["sorv"]:
        rp, ci, v = np.array(k["row_ptr"], np.int32), np.array(k["col_ind"], np.int32), np.array(k["val"], np.float64)
        A, d = P.Matrix(0, k["n"], k["n"], rp, ci, v), P.Descr()
        x, b = np.array(k["x0"], np.float64), np.array(k["b"], np.float64)
        assert L.aoclsparse_dsorv(0, d.h, A.h, k["omega"], 1.0, P._ptr(x), P._ptr(b)) == 0
        assert np.allclose(x, k["x_iter1"], rtol=tol, atol=tol)
        if "x_iter10" in k:
            for _ in range(9):
                assert L.aoclsparse_dsorv(0, d.h, A.h, k["omega"], 1.0, P._ptr(x), P._ptr(b)) == 0
            assert np.allclose(x, k["x_iter10"], rtol=tol, atol=tol)
            assert L.aoclsparse_dsorv(0, d.h, A.h, k["omega"], 0.0, P._ptr(x), P._ptr(b)) == 0
            assert np.allclose(x, k["x_iter10_then_alpha0"], rtol=tol, atol=tol)
    for base in (0, 1):
        n = 6000
        rp, ci, v = random_csr(700 + base, n, n, lambda r, i: r.integers(0, 9), base=base, sort=False)
        # add a dominant diagonal where missing / replace where present: rebuild rows with exactly one diagonal entry
        rows = []
        for i in range(n):
            c, w = ci[rp[i] - base:rp[i + 1] - base] - base, v[rp[i] - base:rp[i + 1] - base]
            keep = c != i
            c, w = np.append(c[keep], i), np.append(w[keep], 10.0 + i % 3)
            perm = np.random.default_rng(i).permutation(len(c))
            rows.append((c[perm], w[perm]))
        rp2 = np.zeros(n + 1, np.int64)
        rp2[1:] = np.cumsum([len(c) for c, _ in rows])
        ci2 = (np.concatenate([c for c, _ in rows]) + base).astype(np.int32)
        v2 = np.concatenate([w for _, w in rows])
        rp2 = (rp2 + base).astype(np.int32)
        A, d = P.Matrix(base, n, n, rp2, ci2, v2), P.Descr(base=base)
        rng = np.random.default_rng(3)
        x0, b = rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)
        for omega, alpha in ((1.0, 1.0), (0.7, -0.5), (1.3, 0.0)):
            st, want = oracle.dsorv(n, base, rp2, ci2, v2, omega, alpha, x0, b)
            x = x0.copy()
            assert st == 0 and L.aoclsparse_dsorv(0, d.h, A.h, omega, alpha, P._ptr(x), P._ptr(b)) == 0
            assert np.array_equal(x, want), (base, omega, alpha)
            xd, bd = dev(x0), dev(b)
            assert L.aoclsparse_dsorv(0, d.h, A.h, omega, alpha, ctypes.c_void_p(xd.data_ptr()), ctypes.c_void_p(bd.data_ptr())) == 0
            torch.cuda.synchronize()
            assert np.array_equal(xd.cpu().numpy(), want)
        x = np.zeros(n)
        for _ in range(25):
            assert L.aoclsparse_dsorv(0, d.h, A.h, 1.0, 1.0, P._ptr(x), P._ptr(b)) == 0
        D = np.zeros((n, n))
        for i in range(n):
            D[i, ci2[rp2[i] - base:rp2[i + 1] - base] - base] = v2[rp2[i] - base:rp2[i + 1] - base]
        assert np.linalg.norm(D @ x - b) <= 1e-10 * np.linalg.norm(b)
        xf, bf = x0.astype(np.float32), b.astype(np.float32)
        Af = P.Matrix(base, n, n, rp2, ci2, v2.astype(np.float32))
        assert L.aoclsparse_ssorv(0, d.h, Af.h, 0.7, 1.0, P._ptr(xf), P._ptr(bf)) == 0
        st, want = oracle.dsorv(n, base, rp2, ci2, v2, 0.7, 1.0, x0, b)
        assert np.allclose(xf, want, rtol=0, atol=64 * EPS32 * max(1.0, np.abs(want).max()))
