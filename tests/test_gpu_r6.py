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
