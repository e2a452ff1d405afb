"""Round-6 GPU tests.  Merge-path SpMV without cross-launch state: a captured launch replayed with a different x every time
(round 5's epoch tag was frozen by the capture), and the same handle under out-of-order residency (pieces meet through
arrival counters, nobody waits)."""
import ctypes
import os
import sys

import numpy as np
import pytest

import oracle
from util import EPS64, abs_row_sums, pkg, random_csr

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import standins  # noqa: E402

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


# --------------------------------------------------------------------------------------------------
# blocked-ELL MFMA csrmm: which XCD works through which block rows (BellPlan::order)
# --------------------------------------------------------------------------------------------------
def _bell_handle(rp, ci, v, m, forced):
    """mm hint + optimize with AOCLSPARSE_MI355_BELL_XCD_CHUNK (read at analysis time) forced / unset"""
    old = os.environ.pop("AOCLSPARSE_MI355_BELL_XCD_CHUNK", None)
    if forced is not None:
        os.environ["AOCLSPARSE_MI355_BELL_XCD_CHUNK"] = str(forced)
    try:
        A = P.Matrix(0, m, m, rp, ci, v)
        d = P.Descr()
        assert L.aoclsparse_set_mm_hint(A.h, P.OP_NONE, d.h, 10) == 0 and L.aoclsparse_optimize(A.h) == 0
    finally:
        os.environ.pop("AOCLSPARSE_MI355_BELL_XCD_CHUNK", None)
        if old is not None:
            os.environ["AOCLSPARSE_MI355_BELL_XCD_CHUNK"] = old
    assert A.spmv_info().mm_bell_width > 0, "the blocked-ELL copy was not built"
    return A, d


def _same_bits(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.int64), np.ascontiguousarray(b).view(np.int64))


@pytest.mark.parametrize("dims,keep,forced", [((12, 16, 16), 1.0, -1), ((12, 16, 16), 1.0, 3), ((12, 16, 16), 1.0, 0), ((16, 24, 20), 0.75, -1),
                                              ((8, 16, 27), 1.0, -1), ((8, 16, 27), 0.75, 5), ((40, 1, 24), 1.0, -1), ((9, 14, 13), 1.0, None)])
def test_blocked_ell_block_row_order_over_the_xcds_same_bits(dims, keep, forced):
    """The order in which the XCDs work through the block rows is a plan of its own (chunks dealt in turn, or a structured grid's
    regions followed through the planes: csrmm_api.cpp choose_bell_order).  Every block row is still computed exactly once by the same
    chain: oracle.dcsrmm's bits (csrmm.hpp:36-90) in every order -- forced chunks that do not divide the block rows, the lattice sweep
    with lines cut into equal and unequal pieces (16, 20: 4 / 5 block rows; 27: 6, 7, 7, 7), with the plane range cut in two
    (16 x 24 x 20: 12 regions per segment), on a 2-D grid (one line per plane), launch order without a list, and whatever the model
    picks for a grid too small to go round the XCDs evenly (9 x 14 x 13); both layouts, 1 / 2 / 4 / 8 wavefronts per block row and a
    column count that is no whole tile, both beta = 0 modes and beta != 0."""
    m, rp, ci, v = standins.block_dense(*dims, keep=keep, seed=31)
    A, d = _bell_handle(rp, ci, v, m, forced)
    info = A.spmv_info()
    if forced == -1:
        # z runs fastest in the stand-in's numbering: the line is dims[2] block rows, a plane dims[1] lines
        assert (info.mm_bell_xcd_chunk, info.mm_bell_lattice_line) == (0, dims[2]), (info.mm_bell_xcd_chunk, info.mm_bell_lattice_line)
        assert info.mm_bell_lattice_lines == dims[1]
        assert info.mm_bell_region_a * info.mm_bell_region_b >= 1
    elif forced is None:
        assert info.mm_bell_xcd_chunk >= 0
    elif forced == 0:
        assert (info.mm_bell_xcd_chunk, info.mm_bell_lattice_line) == (1, 0)
    else:
        assert (info.mm_bell_xcd_chunk, info.mm_bell_lattice_line) == (forced, 0)
    rng = np.random.default_rng(8)
    L.aoclsparse_mi355_set_pointer_mode(P.PTR_DEVICE)
    try:
        for order, n, alpha, beta in ((P.ORDER_ROW, 256, 1.0, 0.0), (P.ORDER_ROW, 64, -0.5, 1.25), (P.ORDER_ROW, 40, 2.0, 0.0),
                                      (P.ORDER_ROW, 130, 1.0, 0.0), (P.ORDER_COLUMN, 64, 1.0, 0.0), (P.ORDER_COLUMN, 21, 1.5, 0.5)):
            B, C0 = rng.uniform(-1, 1, m * n), rng.uniform(-1, 1, m * n)
            if order == P.ORDER_ROW:
                Bc = np.ascontiguousarray(B.reshape(m, n).T).ravel()
                Cc = np.ascontiguousarray(C0.reshape(m, n).T).ravel()
            else:
                Bc, Cc = B, C0
            so, Cr = oracle.dcsrmm("col", alpha, 0, v, ci, rp, m, Bc, n, m, beta, Cc, m)
            assert so == 0
            ref = Cr.reshape(n, m).T if order == P.ORDER_ROW else Cr
            ld = n if order == P.ORDER_ROW else m
            for overwrite in ((False, True) if beta == 0.0 else (False,)):
                Cd = dev(C0)
                assert L.aoclsparse_mi355_set_csrmm_beta0_overwrite(1 if overwrite else 0) == 0
                try:
                    assert P.dcsrmm(P.OP_NONE, alpha, A, d, order, dev(B), n, ld, beta, Cd, ld) == 0
                    torch.cuda.synchronize()
                finally:
                    assert L.aoclsparse_mi355_set_csrmm_beta0_overwrite(0) == 0
                got = Cd.cpu().numpy().reshape(m, n) if order == P.ORDER_ROW else Cd.cpu().numpy()
                assert _same_bits(got, ref), (dims, keep, forced, order, n, alpha, beta, overwrite)
        if forced == 3:
            # new values in place: the blocked copy AND its order list are rebuilt (same structure, same order), the product follows
            v2 = v * rng.uniform(0.5, 1.5, len(v))
            assert L.aoclsparse_dupdate_values(A.h, len(v2), P._ptr(v2)) == 0
            n = 64
            B, C0 = rng.uniform(-1, 1, m * n), rng.uniform(-1, 1, m * n)
            so, Cr = oracle.dcsrmm("col", 1.0, 0, v2, ci, rp, m, np.ascontiguousarray(B.reshape(m, n).T).ravel(), n, m, 0.0,
                                   np.ascontiguousarray(C0.reshape(m, n).T).ravel(), m)
            assert so == 0
            Cd = dev(C0)
            os.environ["AOCLSPARSE_MI355_BELL_XCD_CHUNK"] = "3"
            try:
                assert P.dcsrmm(P.OP_NONE, 1.0, A, d, P.ORDER_ROW, dev(B), n, n, 0.0, Cd, n) == 0
                torch.cuda.synchronize()
            finally:
                os.environ.pop("AOCLSPARSE_MI355_BELL_XCD_CHUNK", None)
            assert A.spmv_info().mm_bell_xcd_chunk == 3
            assert _same_bits(Cd.cpu().numpy().reshape(m, n), Cr.reshape(n, m).T)
    finally:
        L.aoclsparse_mi355_set_pointer_mode(P.PTR_AUTO)


def test_blocked_ell_order_is_chosen_by_the_model_and_travels_with_the_state():
    """32 x 32 nodes per plane, 8 planes (131,072 rows): a plane of 1,024 block rows is more than an XCD keeps in its L2, so launch
    order fetches every stretch of B about five times (the model says so) and the lattice sweep -- 4 x 10 block rows of the
    cross-section followed through the planes -- under two: optimize picks it unasked.  A matrix whose block columns sit at no
    constant offsets (the same grid with its nodes renumbered at random) gets no lattice.  The list is part of the exported
    state: an adopted handle has it and returns the same bits (64 columns against oracle.dcsrmm)."""
    from aocl_sparse_amd.sharded import _DeviceView
    m, rp, ci, v = standins.block_dense(8, 32, 32, seed=41)
    A, d = _bell_handle(rp, ci, v, m, None)
    info = A.spmv_info()
    assert (info.mm_bell_xcd_chunk, info.mm_bell_lattice_line, info.mm_bell_lattice_lines) == (0, 32, 32)
    assert info.mm_bell_model_fetches_launch_order_permille > 4000 and info.mm_bell_model_fetches_permille < 2000
    st, state, ptrs = A.mm_state_export()
    assert st == 0 and state.bytes[12] > 0 and state.bytes[13] == 0
    held = [torch.as_tensor(_DeviceView(p, n), device="cuda").clone() if n else None for p, n in zip(ptrs, list(state.bytes))]
    torch.cuda.synchronize()
    st, R = P.Matrix.mm_state_adopt(state, [t.data_ptr() if t is not None else None for t in held])
    assert st == 0
    del held
    ir = R.spmv_info()
    assert (ir.mm_bell_width, ir.mm_bell_xcd_chunk) == (info.mm_bell_width, 0)
    # a truncated order list is refused
    import ctypes as ct
    bad = P.MmState()
    ct.memmove(ct.addressof(bad), bytes(state), ct.sizeof(bad))
    bad.bytes[12] //= 2
    held = [torch.as_tensor(_DeviceView(p, n), device="cuda").clone() if n else None for p, n in zip(ptrs, list(state.bytes))]
    st, none = P.Matrix.mm_state_adopt(bad, [t.data_ptr() if t is not None else None for t in held])
    assert st == 5 and none is None
    del held
    n = 64
    rng = np.random.default_rng(3)
    B, C0 = rng.uniform(-1, 1, m * n), rng.uniform(-1, 1, m * n)
    so, Cr = oracle.dcsrmm("col", 1.0, 0, v, ci, rp, m, np.ascontiguousarray(B.reshape(m, n).T).ravel(), n, m, 0.0,
                           np.ascontiguousarray(C0.reshape(m, n).T).ravel(), m)
    assert so == 0
    for H in (A, R):
        Cd = dev(C0)
        assert P.dcsrmm(P.OP_NONE, 1.0, H, d, P.ORDER_ROW, dev(B), n, n, 0.0, Cd, n) == 0
        torch.cuda.synchronize()
        assert _same_bits(Cd.cpu().numpy().reshape(m, n), Cr.reshape(n, m).T)
    # the same grid, nodes renumbered at random: no constant offsets, no lattice (whatever chunk the model keeps, the bits stay)
    nodes = m // 16
    perm = np.random.default_rng(1).permutation(nodes)
    rowp = (perm[:, None] * 16 + np.arange(16)[None, :]).ravel()  # new index of old row i
    inv = np.empty(m, dtype=np.int64)
    inv[rowp] = np.arange(m)
    lens = np.diff(rp)[inv]
    rp2 = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    ci2 = np.empty_like(ci)
    v2 = np.empty_like(v)
    for i in range(m):
        a, b, o = rp[inv[i]], rp[inv[i] + 1], rp2[i]
        c = rowp[ci[a:b]]
        s = np.argsort(c, kind="stable")
        ci2[o:o + b - a], v2[o:o + b - a] = c[s], v[a:b][s]
    A2, d2 = _bell_handle(rp2, ci2, v2, m, None)
    i2 = A2.spmv_info()
    assert i2.mm_bell_lattice_line == 0 and i2.mm_bell_xcd_chunk >= 1
    so, Cr2 = oracle.dcsrmm("col", 1.0, 0, v2, ci2, rp2, m, np.ascontiguousarray(B.reshape(m, n).T).ravel(), n, m, 0.0,
                            np.ascontiguousarray(C0.reshape(m, n).T).ravel(), m)
    Cd = dev(C0)
    assert P.dcsrmm(P.OP_NONE, 1.0, A2, d2, P.ORDER_ROW, dev(B), n, n, 0.0, Cd, n) == 0
    torch.cuda.synchronize()
    assert _same_bits(Cd.cpu().numpy().reshape(m, n), Cr2.reshape(n, m).T)


def test_banded_matrix_row_major_csrmm_orders_same_bits_and_travel_with_the_state():
    """A banded matrix (5-point Laplacian, 300 rows per line): the row-per-wavefront kernel deals chunks of band / 8 rows to the XCDs
    (n >= 128, C read) and the narrow kernel walks row blocks that follow the lines, 8 per line (n < 128) -- both only decide WHERE a
    row is computed: oracle.dcsrmm's bits (csrmm.hpp:36-90) for 32, 48, 128 and 256 columns, both beta = 0 modes and beta != 0.  The
    line blocks are the 14th buffer of the exported state: an adopted handle (no analysis of its own) returns the same bits."""
    from aocl_sparse_amd.sharded import _DeviceView
    from util import laplace5
    g = 300
    m, rp, ci, v = laplace5(g)
    A = P.Matrix(0, m, m, rp, ci, v)
    d = P.Descr()
    assert L.aoclsparse_set_mm_hint(A.h, P.OP_NONE, d.h, 10) == 0 and L.aoclsparse_optimize(A.h) == 0
    st, state, ptrs = A.mm_state_export()
    assert st == 0 and state.bytes[13] == 4 * 2 * (8 * g + 1)  # 8 blocks per line of 300 rows + the terminal entry
    held = [torch.as_tensor(_DeviceView(p, n), device="cuda").clone() if n else None for p, n in zip(ptrs, list(state.bytes))]
    torch.cuda.synchronize()
    st, R = P.Matrix.mm_state_adopt(state, [t.data_ptr() if t is not None else None for t in held])
    assert st == 0
    del held
    rng = np.random.default_rng(12)
    L.aoclsparse_mi355_set_pointer_mode(P.PTR_DEVICE)
    try:
        for n, alpha, beta in ((32, 1.0, 0.0), (48, -0.5, 1.25), (128, 1.0, 0.0), (256, 2.0, 0.0), (256, 1.0, -1.0)):
            B, C0 = rng.uniform(-1, 1, m * n), rng.uniform(-1, 1, m * n)
            so, Cr = oracle.dcsrmm("col", alpha, 0, v, ci, rp, m, np.ascontiguousarray(B.reshape(m, n).T).ravel(), n, m, beta,
                                   np.ascontiguousarray(C0.reshape(m, n).T).ravel(), m)
            assert so == 0
            ref = Cr.reshape(n, m).T
            for H in (A, R):
                for overwrite in ((False, True) if beta == 0.0 else (False,)):
                    Cd = dev(C0)
                    assert L.aoclsparse_mi355_set_csrmm_beta0_overwrite(1 if overwrite else 0) == 0
                    try:
                        assert P.dcsrmm(P.OP_NONE, alpha, H, d, P.ORDER_ROW, dev(B), n, n, beta, Cd, n) == 0
                        torch.cuda.synchronize()
                    finally:
                        assert L.aoclsparse_mi355_set_csrmm_beta0_overwrite(0) == 0
                    assert _same_bits(Cd.cpu().numpy().reshape(m, n), ref), (n, alpha, beta, overwrite, H is R)
    finally:
        L.aoclsparse_mi355_set_pointer_mode(P.PTR_AUTO)


@pytest.mark.parametrize("which", ["shell", "flan"])
def test_row_group_csrmm_with_the_groups_band_dealt_to_the_xcds_same_bits(which):
    """Mesh matrices with row groups whose band is wide enough to be dealt (shell: 120 nodes of 5 unknowns per line -> 609 rows;
    flan: 12 x 12 nodes of 3 unknowns per plane -> 473 rows): csrmm_rowgroup2_kernel (n >= 128) and csrmm_rowgroup_sub_kernel
    (64 and 32 columns) in the dealt order return oracle.dcsrmm's bits (csrmm.hpp:36-90), both beta = 0 modes and beta != 0."""
    if which == "shell":
        m, rp, ci, v = standins.shell_like(n=5 * 120 * 60, width=120)
    else:
        m, rp, ci, v = standins.flan_like(nx=12, ny=12, nz=12)
    A = P.Matrix(0, m, m, rp, ci, v)
    d = P.Descr()
    assert L.aoclsparse_set_mm_hint(A.h, P.OP_NONE, d.h, 10) == 0 and L.aoclsparse_optimize(A.h) == 0
    rng = np.random.default_rng(17)
    L.aoclsparse_mi355_set_pointer_mode(P.PTR_DEVICE)
    try:
        for n, alpha, beta in ((256, 1.0, 0.0), (130, -0.5, 1.25), (64, 2.0, 0.0), (32, 1.0, 0.0)):
            B, C0 = rng.uniform(-1, 1, m * n), rng.uniform(-1, 1, m * n)
            so, Cr = oracle.dcsrmm("col", alpha, 0, v, ci, rp, m, np.ascontiguousarray(B.reshape(m, n).T).ravel(), n, m, beta,
                                   np.ascontiguousarray(C0.reshape(m, n).T).ravel(), m)
            assert so == 0
            ref = Cr.reshape(n, m).T
            for overwrite in ((False, True) if beta == 0.0 else (False,)):
                Cd = dev(C0)
                assert L.aoclsparse_mi355_set_csrmm_beta0_overwrite(1 if overwrite else 0) == 0
                try:
                    assert P.dcsrmm(P.OP_NONE, alpha, A, d, P.ORDER_ROW, dev(B), n, n, beta, Cd, n) == 0
                    torch.cuda.synchronize()
                finally:
                    assert L.aoclsparse_mi355_set_csrmm_beta0_overwrite(0) == 0
                assert A.spmv_info().mm_groups > 0
                assert _same_bits(Cd.cpu().numpy().reshape(m, n), ref), (which, n, alpha, beta, overwrite)
    finally:
        L.aoclsparse_mi355_set_pointer_mode(P.PTR_AUTO)


def test_lane_per_block_trsv_on_an_irregular_numbering_narrow_slices_and_sorted_levels():
    """The shell-like mesh with a tenth of its couplings dropped and its nodes renumbered at random inside windows of 256 (a small
    instance of the unstructured stand-in): 5-row blocks on up to 20 dependencies, slices that wait for many producer slices -- the plan
    then packs 32 blocks per wavefront and orders the blocks of a level by their last dependency (trsv_api.cpp).  None of that may touch
    a row's chain: ref_trsv_*'s bits (trsv_kr.hpp:57-75) on every triangle, unit and non-unit, on the lane-per-block schedule, on the
    automatic one, and with a pinned KT kid."""
    from test_gpu_trsv_blocks import VARIANTS
    from util import kt_lanes, trsv_schedule
    m, rp, ci, v = standins.shell_like_unstructured(n=5 * 24000, width=120)
    # ILU(0)-like triangles: the matrix itself, diagonally dominant enough for a stable solve
    rid = np.repeat(np.arange(m), np.diff(rp))
    v = v.copy()
    v[ci == rid] = np.sign(v[ci == rid]) * (np.abs(v[ci == rid]) + 40.0)
    o = oracle.dcsr_optimize(m, m, len(v), 0, rp, ci, v)
    A = P.Matrix(0, m, m, rp, ci, v)
    rng = np.random.default_rng(23)
    for kind, fill, op in VARIANTS:
        for unit in (True, False):
            d = P.Descr(mtype=P.TYPE_TRIANGULAR, fill=getattr(P, fill), diag=P.DIAG_UNIT if unit else P.DIAG_NON_UNIT)
            b = rng.uniform(-1, 1, m)
            iend = o["idiag"] if kind[0] == "l" else o["iurow"]
            st, xr = oracle.dtrsv(kind, 0.75, m, 0, o["val"], o["ind"], o["ptr"], iend, b, unit)
            assert st == 0
            for sched in (4, -1):
                with trsv_schedule(P, sched):
                    xd = torch.full((m,), 7.0, dtype=torch.float64, device="cuda")
                    assert P.dtrsv(getattr(P, op), 0.75, A, d, dev(b), xd) == 0
                    torch.cuda.synchronize()
                assert np.array_equal(xd.cpu().numpy(), xr), (kind, unit, sched)
        info = A.trsv_info(getattr(P, fill), getattr(P, op))
        assert info.blocks > 0 and info.block_levels > 100, (kind, info.blocks, info.block_levels)
        # many producers per slice -> 32 blocks per wavefront: more slices than blocks / 64 would give
        assert info.slice_fan_in_permille > 8000 and info.slices > info.blocks / 40, (kind, info.slice_fan_in_permille, info.slices, info.blocks)
    # kid 3 on the L solve: the KT form of the block kernel on the same plan
    d = P.Descr(mtype=P.TYPE_TRIANGULAR, fill=P.FILL_LOWER, diag=P.DIAG_NON_UNIT)
    b = rng.uniform(-1, 1, m)
    st, xk = oracle.trsv_kt("l", kt_lanes(3, np.float64), 0.75, m, 0, o["val"], o["ind"], o["ptr"], o["idiag"], b, False)
    assert st == 0
    xd = torch.full((m,), 7.0, dtype=torch.float64, device="cuda")
    assert P.dtrsv(P.OP_NONE, 0.75, A, d, dev(b), xd, kid=3) == 0
    torch.cuda.synchronize()
    assert np.array_equal(xd.cpu().numpy().view(np.uint64), xk.view(np.uint64))


def test_multi_device_replicas_carry_the_order_plans():
    """aoclsparse_mi355_dcsrmm_multi with three slots on device 0: the replicas are device-to-device clones of the primary's state --
    including the blocked-ELL order list (forced chunks of 3 block rows) and, for a banded matrix, the row blocks along its lines
    (13-column slabs: the narrow kernel).  Every slot's slab equals the single-call product bit for bit, also after a value update."""
    st, dev0, _, _ = P.device_info()
    assert st == 0
    rng = np.random.default_rng(77)
    m, rp, ci, v = standins.block_dense(12, 16, 16, seed=5)
    A, d = _bell_handle(rp, ci, v, m, 3)
    assert A.spmv_info().mm_bell_xcd_chunk == 3
    from util import laplace5
    ml, rpl, cil, vl = laplace5(300)
    Al = P.Matrix(0, ml, ml, rpl, cil, vl)
    assert L.aoclsparse_set_mm_hint(Al.h, P.OP_NONE, d.h, 10) == 0 and L.aoclsparse_optimize(Al.h) == 0
    for H, mm, vals, n in ((A, m, v, 192), (Al, ml, vl, 40)):
        B, C0 = rng.uniform(-1, 1, mm * n), rng.uniform(-1, 1, mm * n)
        for rnd in range(2):
            ref, C = C0.copy(), C0.copy()
            assert P.dcsrmm(P.OP_NONE, 1.25, H, d, P.ORDER_ROW, B, n, n, -0.5, ref, n) == 0
            assert P.dcsrmm_multi(P.OP_NONE, 1.25, H, d, P.ORDER_ROW, B, n, n, -0.5, C, n, [dev0] * 3) == 0
            assert np.array_equal(C, ref), (mm, rnd)
            if rnd == 0:
                assert L.aoclsparse_mi355_replicas_cloned(H.h) == 2
                v2 = np.ascontiguousarray(vals * rng.uniform(0.5, 1.5, len(vals)))
                os.environ["AOCLSPARSE_MI355_BELL_XCD_CHUNK"] = "3"
                try:
                    assert L.aoclsparse_dupdate_values(H.h, len(v2), P._ptr(v2)) == 0
                    # (the next products rebuild the value-holding copies, on the primary and on fresh replicas)
                    chk = C0.copy()
                    assert P.dcsrmm(P.OP_NONE, 1.25, H, d, P.ORDER_ROW, B, n, n, -0.5, chk, n) == 0
                finally:
                    os.environ.pop("AOCLSPARSE_MI355_BELL_XCD_CHUNK", None)
