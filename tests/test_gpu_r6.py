"""Round-6 GPU tests.  Merge-path SpMV without cross-launch state: a captured launch replayed with a different x every time
(round 5's epoch tag was frozen by the capture), and the same handle under out-of-order residency (pieces meet through
arrival counters, nobody waits)."""
import ctypes

import numpy as np
import pytest

import oracle
from util import EPS64, abs_row_sums, pkg, random_csr

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
P = pkg()
L = P.lib()


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    assert torch.cuda.is_available(), "GPU tests need a GPU (no CPU fallback exists)"
    yield
    L.aoclsparse_mi355_set_pointer_mode(P.PTR_AUTO)


def _merge_handle(seed, m, long_rows, length):
    """tridiagonal-ish rows of 0..7 entries + a few rows of `length` entries (cut by many 1,024-item tiles), forced onto merge-path"""
    rp, ci, v = random_csr(seed, m, m, lambda r, i: length if i in long_rows else r.integers(0, 8))
    assert L.aoclsparse_mi355_set_option(P.OPTION_SPMV_KERNEL, 2) == 0
    assert L.aoclsparse_mi355_set_option(P.OPTION_SELL, 0) == 0
    try:
        A = P.Matrix(0, m, m, rp, ci, v)
        d = P.Descr()
        assert L.aoclsparse_set_mv_hint(A.h, P.OP_NONE, d.h, 100) == 0 and L.aoclsparse_optimize(A.h) == 0
        assert A.spmv_info().kernel == 2
    finally:
        assert L.aoclsparse_mi355_set_option(P.OPTION_SPMV_KERNEL, 0) == 0
        assert L.aoclsparse_mi355_set_option(P.OPTION_SELL, -1) == 0
    return A, d, rp, ci, v


def test_merge_path_launch_replayed_from_a_hip_graph_with_a_new_x_every_time():
    """A forced merge-path handle (rows of 9,000 entries: each crosses ~9 tiles), one aoclsparse_dmv captured into a HIP graph and
    replayed six times; x is rewritten in place between replays (a solver's iterate).  Every replay must return exactly what an
    eager call on the same x returns -- cut rows included: the pieces of a replay must never be mistaken for those of the
    previous one."""
    m = 30000
    A, d, rp, ci, v = _merge_handle(77, m, (3, 14000, 29990), 9000)
    rng = np.random.default_rng(5)
    xs = [rng.uniform(-1, 1, m) for _ in range(6)]
    L.aoclsparse_mi355_set_pointer_mode(P.PTR_DEVICE)
    s = torch.cuda.Stream()
    assert L.aoclsparse_mi355_set_stream(ctypes.c_void_p(s.cuda_stream)) == 0
    try:
        with torch.cuda.stream(s):
            x = dev(xs[0])
            y = torch.zeros(m, dtype=torch.float64, device="cuda")
            eager = []
            for k in range(6):  # (also creates this stream's piece set)
                x.copy_(dev(xs[k]))
                y.fill_(float("nan"))
                assert P.dmv(P.OP_NONE, 1.0, A, d, x, 0.0, y) == 0
                s.synchronize()
                eager.append(y.cpu().numpy().copy())
                so, yr = oracle.dcsrmv(-1, 0, 1.0, m, len(v), v, ci, rp, xs[k], 0.0, np.zeros(m))
                scale = abs_row_sums(rp, ci, v, xs[k])
                assert np.all(np.abs(eager[k] - yr) <= (np.diff(rp) + 24) * EPS64 * scale + 1e-300)
            assert not np.array_equal(eager[0], eager[1])
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                assert P.dmv(P.OP_NONE, 1.0, A, d, x, 0.0, y) == 0
            for k in (3, 0, 5, 1, 4, 2, 2):
                x.copy_(dev(xs[k]))
                y.fill_(float("nan"))
                g.replay()
                s.synchronize()
                got = y.cpu().numpy()
                assert np.array_equal(got, eager[k]), (k, int(np.sum(got != eager[k])))
    finally:
        assert L.aoclsparse_mi355_set_stream(None) == 0
        L.aoclsparse_mi355_set_pointer_mode(P.PTR_AUTO)


def test_merge_path_float_and_base_one_through_the_arrival_counters():
    """the same protocol for float values (4-byte pieces) and one-based indices; β ≠ 0 on cut rows"""
    m = 12000
    rp, ci, v = random_csr(9, m, m, lambda r, i: 5000 if i in (0, 6000, 11999) else r.integers(0, 6), base=1, dtype=np.float32)
    assert L.aoclsparse_mi355_set_option(P.OPTION_SPMV_KERNEL, 2) == 0
    assert L.aoclsparse_mi355_set_option(P.OPTION_SELL, 0) == 0
    try:
        A = P.Matrix(1, m, m, rp, ci, v)
        d = P.Descr(base=1)
        assert L.aoclsparse_set_mv_hint(A.h, P.OP_NONE, d.h, 100) == 0 and L.aoclsparse_optimize(A.h) == 0
    finally:
        assert L.aoclsparse_mi355_set_option(P.OPTION_SPMV_KERNEL, 0) == 0
        assert L.aoclsparse_mi355_set_option(P.OPTION_SELL, -1) == 0
    x = np.random.default_rng(1).uniform(-1, 1, m).astype(np.float32)
    y0 = np.random.default_rng(2).uniform(-1, 1, m).astype(np.float32)
    outs = []
    for _ in range(3):
        yd = dev(y0)
        assert P.smv(P.OP_NONE, 1.5, A, d, dev(x), -0.25, yd) == 0
        torch.cuda.synchronize()
        outs.append(yd.cpu().numpy())
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    rp0, ci0 = rp.astype(np.int64) - 1, ci.astype(np.int64) - 1
    exact = 1.5 * np.add.reduceat(np.concatenate([v.astype(np.float64) * x[ci0], [0.0]]), np.minimum(rp0[:-1], len(v))) * (np.diff(rp0) > 0) \
        - 0.25 * y0
    scale = np.add.reduceat(np.concatenate([np.abs(v.astype(np.float64) * x[ci0]), [0.0]]), np.minimum(rp0[:-1], len(v))) * (np.diff(rp0) > 0)
    eps32 = 2.0 ** -23
    assert np.all(np.abs(outs[0] - exact) <= (np.diff(rp0) + 24) * eps32 * (1.5 * scale + np.abs(y0)) + 1e-30)


# --------------------------------------------------------------------------------------------------
# the two-level TRSV schedule (schedule 5: chunks of consecutive blocks, hand-offs inside a chunk through LDS)
# --------------------------------------------------------------------------------------------------
from test_gpu_trsv_blocks import VARIANTS, fixed, mixed, node_mesh  # noqa: E402
from util import trsv_schedule  # noqa: E402


@pytest.fixture
def forced_chunks():
    """aoclsparse_mi355_set_option(trsv_chunks, 1): build the chunk plan whatever the plan-time model says (the meshes here are small)"""
    assert L.aoclsparse_mi355_set_option(P.OPTION_TRSV_CHUNKS, 1) == 0
    yield
    assert L.aoclsparse_mi355_set_option(P.OPTION_TRSV_CHUNKS, -1) == 0


@pytest.mark.parametrize("name,dofs,width,far", [("five", fixed(5), 37, 0), ("two", fixed(2), 50, 0), ("eight", fixed(8), 29, 0),
                                                 ("three+long", fixed(3), 41, 30), ("mixed", mixed, 33, 0),
                                                 ("mixed+long", mixed, 64, 28)])
def test_two_level_trsv_bit_exact_every_triangle(forced_chunks, name, dofs, width, far):
    """L, L^T, U^T and U (whose rows START with the rows of their own block: the block's rows are phases one after the other), unit
    and non-unit, blocks of 1-8 rows, single rows with more dependencies than a step polls in one batch, several chunks with halos: x must be the
    serial chain of ref_trsv_* bit for bit (trsv_kr.hpp:57-75), and the schedule that ran must be the two-level one."""
    nodes = 12000
    m, rp, ci, v = node_mesh(900 + len(name), nodes, width, dofs(np.random.default_rng(1), nodes), far=far)
    o = oracle.dcsr_optimize(m, m, len(v), 0, rp, ci, v)
    A = P.Matrix(0, m, m, rp, ci, v)
    rng = np.random.default_rng(11)
    ran = 0
    for kind, fill, op in VARIANTS:
        for unit in (True, False):
            d = P.Descr(mtype=P.TYPE_TRIANGULAR, fill=getattr(P, fill), diag=P.DIAG_UNIT if unit else P.DIAG_NON_UNIT)
            b = rng.uniform(-1, 1, m)
            st, xr = oracle.dtrsv(kind, 0.75, m, 0, o["val"], o["ind"], o["ptr"], o["idiag"] if kind[0] == "l" else o["iurow"], b, unit)
            assert st == 0
            with trsv_schedule(P, 5):
                xd = torch.full((m,), 7.0, dtype=torch.float64, device="cuda")
                assert P.dtrsv(getattr(P, op), 0.75, A, d, dev(b), xd) == 0
                torch.cuda.synchronize()
                info = A.trsv_info(getattr(P, fill), getattr(P, op))
            got = xd.cpu().numpy()
            assert np.array_equal(got, xr), (name, kind, unit, int((got != xr).sum()))
            assert info.schedule == 5 and info.chunks >= 2 and info.steps > info.chunks, (name, kind, info.schedule, info.chunks)
            ran += 1
    assert ran == 8


def test_two_level_trsv_float_strided_and_trsm(forced_chunks):
    """float values (4-byte tagged words), strided b / x, and several right-hand sides (one grid column per right-hand side, own
    ticket and solution slab) on the two-level schedule"""
    nodes = 9000
    m, rp, ci, v = node_mesh(77, nodes, 45, mixed(np.random.default_rng(3), nodes), far=5)
    o = oracle.dcsr_optimize(m, m, len(v), 0, rp, ci, v)
    # float
    vf = v.astype(np.float32)
    Af = P.Matrix(0, m, m, rp, ci, vf)
    bf = np.random.default_rng(6).uniform(-1, 1, m).astype(np.float32)
    for unit in (True, False):
        d = P.Descr(mtype=P.TYPE_TRIANGULAR, fill=P.FILL_LOWER, diag=P.DIAG_UNIT if unit else P.DIAG_NON_UNIT)
        xo = np.zeros(m, np.float32)
        assert oracle.lib().orc_strsv_l(ctypes.c_float(1.0), m, 0, P._ptr(vf), P._ptr(ci), P._ptr(rp), P._ptr(o["idiag"]), P._ptr(bf), 1,
                                        P._ptr(xo), 1, 1 if unit else 0) == 0
        with trsv_schedule(P, 5):
            xd = torch.zeros(m, dtype=torch.float32, device="cuda")
            assert P.strsv(P.OP_NONE, 1.0, Af, d, dev(bf), xd) == 0
            torch.cuda.synchronize()
            assert Af.trsv_info(P.FILL_LOWER).schedule == 5
        assert np.array_equal(xd.cpu().numpy(), xo), unit
    # strided, double
    A = P.Matrix(0, m, m, rp, ci, v)
    d = P.Descr(mtype=P.TYPE_TRIANGULAR, fill=P.FILL_LOWER)
    incb, incx = 3, 2
    b = np.random.default_rng(9).uniform(-1, 1, m * incb)
    st, xr = oracle.dtrsv("l", 1.5, m, 0, o["val"], o["ind"], o["ptr"], o["idiag"], b[::incb].copy(), False)
    with trsv_schedule(P, 5):
        xs = torch.full((m * incx,), 7.0, dtype=torch.float64, device="cuda")
        assert P.dtrsv(P.OP_NONE, 1.5, A, d, dev(b), xs, incb=incb, incx=incx) == 0
        torch.cuda.synchronize()
    got = xs.cpu().numpy()
    assert np.array_equal(got[::incx], xr) and np.all(got[1::incx] == 7.0)
    # several right-hand sides, both layouts
    n = 5
    rng = np.random.default_rng(12)
    for kind, fill, op, unit in (("l", P.FILL_LOWER, P.OP_NONE, True), ("lt", P.FILL_LOWER, P.OP_TRANSPOSE, False),
                                 ("ut", P.FILL_UPPER, P.OP_TRANSPOSE, True), ("u", P.FILL_UPPER, P.OP_NONE, False)):
        dd = P.Descr(mtype=P.TYPE_TRIANGULAR, fill=fill, diag=P.DIAG_UNIT if unit else P.DIAG_NON_UNIT)
        iend = o["idiag"] if kind[0] == "l" else o["iurow"]
        for lay, shape, col in ((P.ORDER_COLUMN, (n, m + 3), lambda M, j: M[j, :m]), (P.ORDER_ROW, (m, n + 2), lambda M, j: M[:, j])):
            Bm = rng.uniform(-1, 1, shape)
            Xd = dev(np.full(shape, 7.0))
            with trsv_schedule(P, 5):
                assert L.aoclsparse_dtrsm(op, 0.5, A.h, dd.h, lay, P._ptr(dev(Bm)), n, shape[1], P._ptr(Xd), shape[1]) == 0
                torch.cuda.synchronize()
            X = Xd.cpu().numpy()
            for j in range(n):
                st, xr = oracle.dtrsv(kind, 0.5, m, 0, o["val"], o["ind"], o["ptr"], iend, np.ascontiguousarray(col(Bm, j)), unit)
                assert st == 0 and np.array_equal(col(X, j), xr), (kind, lay, j)


def test_two_level_trsv_nan_inf_and_tag_propagate(forced_chunks):
    """NaN, +-Inf and the exact NOT-READY bit pattern in b on the two-level schedule: the same propagation as the serial chain"""
    nodes = 8000
    m, rp, ci, v = node_mesh(33, nodes, 40, mixed(np.random.default_rng(2), nodes))
    o = oracle.dcsr_optimize(m, m, len(v), 0, rp, ci, v)
    A = P.Matrix(0, m, m, rp, ci, v)
    tag = np.array([0x7FF8DEADBEEF0355], dtype=np.uint64).view(np.float64)[0]
    d = P.Descr(mtype=P.TYPE_TRIANGULAR, fill=P.FILL_LOWER)
    b = np.random.default_rng(8).uniform(-1, 1, m)
    b[[0, 40, m // 2]] = [np.nan, np.inf, tag]
    st, xr = oracle.dtrsv("l", 1.0, m, 0, o["val"], o["ind"], o["ptr"], o["idiag"], b, False)
    with trsv_schedule(P, 5):
        xd = torch.zeros(m, dtype=torch.float64, device="cuda")
        assert P.dtrsv(P.OP_NONE, 1.0, A, d, dev(b), xd) == 0
        torch.cuda.synchronize()
        assert A.trsv_info(P.FILL_LOWER).schedule == 5
    got = xd.cpu().numpy()
    gn, rn = np.isnan(got), np.isnan(xr)
    assert np.array_equal(gn, rn) and np.array_equal(got[~gn], xr[~rn]) and 2 <= rn.sum() < m


def test_two_level_trsv_is_chosen_by_the_model_where_the_dag_is_deep_and_narrow():
    """The plan-time model of the triangle's DAG selects the schedule: the shell-like ILU(0) factor (a mesh numbered line by
    line: 1,101 block levels of <= 500 blocks) gets the two-level schedule, its unstructured variant (nodes renumbered at random
    inside windows: a shallow, wide DAG whose chunks would run one after the other) keeps the lane-per-block one.  Both at a
    tenth of the full size here; the full size is test_shell_like_ilu0_trsv_full_size."""
    sys_path = __import__("sys").path
    import os
    sys_path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import standins
    for variant, want in (("shell", 5), ("unstructured", 4)):
        m, rp, ci, v = standins.shell_like(n=150000) if variant == "shell" else standins.shell_like_unstructured(n=150000)
        st, lu, dg = oracle.dilu0(m, 0, rp, ci, v)
        o = oracle.dcsr_optimize(m, m, len(lu), 0, rp, ci, lu)
        A = P.Matrix(0, m, m, rp, ci, lu)
        d = P.Descr(mtype=P.TYPE_TRIANGULAR, fill=P.FILL_LOWER, diag=P.DIAG_UNIT)
        b = np.random.default_rng(2).uniform(-1, 1, m)
        st, xr = oracle.dtrsv("l", 1.0, m, 0, lu, ci, rp, o["idiag"], b, True)
        xd = torch.zeros(m, dtype=torch.float64, device="cuda")
        assert P.dtrsv(P.OP_NONE, 1.0, A, d, dev(b), xd) == 0
        torch.cuda.synchronize()
        info = A.trsv_info(P.FILL_LOWER)
        assert info.schedule == want, (variant, info.schedule, info.model_chunk_us, info.model_block_us)
        assert np.array_equal(xd.cpu().numpy(), xr), variant


def test_two_level_trsv_after_update_values(forced_chunks):
    """aoclsparse_dupdate_values between two solves: the TRSV plans (block plan, chunk plan with its staged copies of the values)
    are dropped and rebuilt from the new values -- the second solve is the serial chain on the NEW matrix, still on schedule 5"""
    nodes = 6000
    m, rp, ci, v = node_mesh(5, nodes, 40, np.full(nodes, 5))
    A = P.Matrix(0, m, m, rp, ci, v)
    d = P.Descr(mtype=P.TYPE_TRIANGULAR, fill=P.FILL_LOWER)
    b = np.random.default_rng(3).uniform(-1, 1, m)
    for rnd in range(2):
        if rnd == 1:
            v2 = v * np.random.default_rng(9).uniform(0.9, 1.1, len(v))
            assert L.aoclsparse_dupdate_values(A.h, len(v2), P._ptr(v2)) == 0
            v = v2
        o = oracle.dcsr_optimize(m, m, len(v), 0, rp, ci, v)
        st, xr = oracle.dtrsv("l", 1.0, m, 0, o["val"], o["ind"], o["ptr"], o["idiag"], b, False)
        with trsv_schedule(P, 5):
            xd = torch.zeros(m, dtype=torch.float64, device="cuda")
            assert P.dtrsv(P.OP_NONE, 1.0, A, d, dev(b), xd) == 0
            torch.cuda.synchronize()
            assert A.trsv_info(P.FILL_LOWER).schedule == 5
        assert np.array_equal(xd.cpu().numpy(), xr), rnd
