"""GPU tier (-m gpu): the supernodal TRSV schedule (one lane per block of chained rows, csrc/trsv_kernels.hip
trsv_block_kernel) against the serial reference chain of the oracle -- all four (fill, op) triangles, both diagonal
types, both real precisions, every compiled shape (blocks of <= 5 rows with <= 16 external dependencies: values in
registers; larger: row by row from LDS), blocks of mixed sizes, rows too long for a block (the kernel's tail loop),
NaN / Inf / NOT-READY-tag propagation.  Bar: bit-exact (same chain order per row).  That the block schedule is the one
that ran is checked through its diagnostic trace in a subprocess (the C ABI is frozen: there is no query for it)."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle
from util import ROOT, pkg

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
P = pkg()
L = P.lib()


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def node_mesh(seed, nodes, width, dofs, keep=0.9, far=0, fourth=False):
    """Sorted CSR of a mesh matrix with dense node-to-node blocks: node i has dofs[i] unknowns and couples to the nodes
    i-1, i-width, i-width-1 (and symmetrically i+1, i+width, i+width+1; `fourth`: also i-width+1 / i+width-1, a nine-point
    stencil: FOUR lower neighbours), each kept with probability `keep`, plus `far` random extra neighbours for every 50th node
    (rows too long for a block).  Strong diagonal, small off-diagonals."""
    rng = np.random.default_rng(seed)
    dofs = np.asarray(dofs, dtype=np.int64)
    first = np.concatenate([[0], np.cumsum(dofs)])
    m = int(first[-1])
    nbr = [set() for _ in range(nodes)]
    for i in range(nodes):
        for d in (1, width, width + 1) + ((width - 1,) if fourth else ()):
            j = i - d
            if (j >= 0 and (d != 1 or i % width) and (d != width + 1 or i % width) and (d != width - 1 or (i + 1) % width)
                    and rng.random() < keep):
                nbr[i].add(j), nbr[j].add(i)
        if far and i % 50 == 25:
            for j in rng.integers(0, nodes, size=far):
                if j != i:
                    nbr[i].add(int(j)), nbr[int(j)].add(i)
    rp, ci = [0], []
    for i in range(nodes):
        cols = np.concatenate([np.arange(first[j], first[j + 1]) for j in sorted(nbr[i] | {i})])
        for _ in range(dofs[i]):
            ci.append(cols)
            rp.append(rp[-1] + len(cols))
    ci = np.concatenate(ci)
    rid = np.repeat(np.arange(m), np.diff(rp))
    val = rng.uniform(-1.0, 1.0, size=len(ci)) / 8.0
    val[ci == rid] = rng.uniform(2.0, 4.0, size=m) * rng.choice([-1.0, 1.0], size=m)
    return m, np.asarray(rp, dtype=np.int32), ci.astype(np.int32), val


def fixed(n):
    return lambda rng, nodes: np.full(nodes, n)


def mixed(rng, nodes):
    return rng.integers(1, 9, size=nodes)


VARIANTS = [("l", "FILL_LOWER", "OP_NONE"), ("u", "FILL_UPPER", "OP_NONE"), ("lt", "FILL_LOWER", "OP_TRANSPOSE"),
            ("ut", "FILL_UPPER", "OP_TRANSPOSE")]


def solve_all(m, rp, ci, v, check):
    o = oracle.dcsr_optimize(m, m, len(v), 0, rp, ci, v)
    A = P.Matrix(0, m, m, rp, ci, v)
    rng = np.random.default_rng(11)
    for kind, fill, op in VARIANTS:
        for unit in (True, False):
            d = P.Descr(mtype=P.TYPE_TRIANGULAR, fill=getattr(P, fill), diag=P.DIAG_UNIT if unit else P.DIAG_NON_UNIT)
            b = rng.uniform(-1, 1, m)
            st, xr = oracle.dtrsv(kind, 0.75, m, 0, o["val"], o["ind"], o["ptr"], o["idiag"] if kind[0] == "l" else o["iurow"],
                                  b, unit)
            assert st == 0
            for kid in (None, 0):  # automatic kid and the pinned reference kernel: the same chain
                xd = torch.full((m,), 7.0, dtype=torch.float64, device="cuda")
                assert P.dtrsv(getattr(P, op), 0.75, A, d, dev(b), xd, kid=kid) == 0
                torch.cuda.synchronize()
                check(xd.cpu().numpy(), xr, (kind, unit, kid))
    return A


@pytest.mark.parametrize("name,dofs,width,far", [("five", fixed(5), 37, 0), ("two", fixed(2), 50, 0), ("eight", fixed(8), 29, 0),
                                                 ("three+long", fixed(3), 41, 12), ("mixed", mixed, 33, 0),
                                                 ("mixed+long", mixed, 64, 9)])
def test_block_trsv_bit_exact_every_triangle(name, dofs, width, far):
    nodes = 3000
    m, rp, ci, v = node_mesh(500 + len(name), nodes, width, dofs(np.random.default_rng(1), nodes), far=far)

    def same(got, ref, what):
        assert np.array_equal(got, ref), (name, what, int((got != ref).sum()))

    solve_all(m, rp, ci, v, same)


def test_block_trsv_five_rows_on_twenty_dependencies():
    """Nodes of 5 unknowns below FOUR neighbours (a nine-point node stencil, every coupling kept): 5-row blocks on 20 external
    dependencies = 110 entries, the largest block the plan forms (TRSV_BLK_NV) and the largest shape whose values stay in registers
    across the wait (round 6; such nodes used to be cut into 4 rows + 1).  Every triangle, unit / non-unit, the automatic kid, the
    pinned reference kid, the lane-per-block and the two-level schedule: ref_trsv_*'s bits; kid 1 / 2 / 3 (trsv_block_kt_kernel, whose
    LDS for 110 entries per lane is requested dynamically): kt_trsv_*'s bits."""
    from util import kt_lanes, trsv_schedule
    nodes = 4000
    m, rp, ci, v = node_mesh(4242, nodes, 40, np.full(nodes, 5), keep=1.0, fourth=True)

    def same(got, ref, what):
        assert np.array_equal(got, ref), (what, int((got != ref).sum()))

    A = solve_all(m, rp, ci, v, same)
    info = A.trsv_info(P.FILL_LOWER, P.OP_NONE)
    # one block per node: 4,000 blocks of 5 rows (not 8,000 of 4 + 1)
    assert info.blocks == nodes, info.blocks
    o = oracle.dcsr_optimize(m, m, len(v), 0, rp, ci, v)
    rng = np.random.default_rng(5)
    assert L.aoclsparse_mi355_set_option(P.OPTION_TRSV_CHUNKS, 1) == 0
    try:
        for kind, fill, op in VARIANTS:
            d = P.Descr(mtype=P.TYPE_TRIANGULAR, fill=getattr(P, fill), diag=P.DIAG_NON_UNIT)
            b = rng.uniform(-1, 1, m)
            iend = o["idiag"] if kind[0] == "l" else o["iurow"]
            st, xr = oracle.dtrsv(kind, 0.75, m, 0, o["val"], o["ind"], o["ptr"], iend, b, False)
            assert st == 0
            for sched in (4, 5):
                with trsv_schedule(P, sched):
                    xd = torch.full((m,), 7.0, dtype=torch.float64, device="cuda")
                    assert P.dtrsv(getattr(P, op), 0.75, A, d, dev(b), xd) == 0
                    torch.cuda.synchronize()
                assert np.array_equal(xd.cpu().numpy(), xr), (kind, sched)
            for kid in (1, 3):
                st, xk = oracle.trsv_kt(kind, kt_lanes(kid, np.float64), 0.75, m, 0, o["val"], o["ind"], o["ptr"], iend, b, False)
                assert st == 0
                xd = torch.full((m,), 7.0, dtype=torch.float64, device="cuda")
                assert P.dtrsv(getattr(P, op), 0.75, A, d, dev(b), xd, kid=kid) == 0
                torch.cuda.synchronize()
                assert np.array_equal(xd.cpu().numpy().view(np.uint64), xk.view(np.uint64)), (kind, kid)
    finally:
        assert L.aoclsparse_mi355_set_option(P.OPTION_TRSV_CHUNKS, -1) == 0


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("name,dofs,width,far", [("five", fixed(5), 37, 0), ("eight", fixed(8), 29, 0),
                                                 ("three+long", fixed(3), 41, 30), ("mixed+long", mixed, 64, 28)])
def test_block_trsv_pinned_kid_kt_orders(name, dofs, width, far, dtype):
    """aoclsparse_?trsv_kid 1 / 2 / 3 on matrices whose triangles have a block plan: trsv_block_kt_kernel (the block kernel's
    protocol with run-time KT loops) must reproduce kt_trsv_l / kt_trsv_u (trsv_kt.cpp:64-150, :297-383) bit for bit -- L and U,
    unit and non-unit, blocks of up to 8 rows, single rows with more dependencies than the kernel polls in one batch (its
    one-by-one path), double and float; the transposed solves (no KT arithmetic of their own) stay on the reference chain."""
    nodes = 2500
    m, rp, ci, v = node_mesh(900 + len(name), nodes, width, dofs(np.random.default_rng(3), nodes), far=far)
    o = oracle.dcsr_optimize(m, m, len(v), 0, rp, ci, v)
    vv = v.astype(dtype)
    ov = o["val"].astype(dtype)
    A = P.Matrix(0, m, m, rp, ci, vv)
    rng = np.random.default_rng(12)
    solve = P.dtrsv if dtype == np.float64 else P.strsv
    u = np.uint64 if dtype == np.float64 else np.uint32
    from util import kt_lanes
    for kind, fill, op in VARIANTS:
        for unit in (True, False):
            d = P.Descr(mtype=P.TYPE_TRIANGULAR, fill=getattr(P, fill), diag=P.DIAG_UNIT if unit else P.DIAG_NON_UNIT)
            b = rng.uniform(-1, 1, m).astype(dtype)
            iend = o["idiag"] if kind[0] == "l" else o["iurow"]
            for kid in (1, 2, 3):
                st, xr = oracle.trsv_kt(kind, kt_lanes(kid, dtype), 0.75, m, 0, ov, o["ind"], o["ptr"], iend, b, unit, dtype=dtype)
                assert st == 0
                xd = torch.full((m,), 7.0, dtype=torch.float64 if dtype == np.float64 else torch.float32, device="cuda")
                assert solve(getattr(P, op), 0.75, A, d, dev(b), xd, kid=kid) == 0
                torch.cuda.synchronize()
                got = xd.cpu().numpy()
                assert np.array_equal(got.view(u), xr.view(u)), (name, kind, unit, kid, int((got != xr).sum()))


def test_block_trsv_strided_and_host_pointers():
    nodes = 1500
    m, rp, ci, v = node_mesh(7, nodes, 30, np.full(nodes, 5))
    o = oracle.dcsr_optimize(m, m, len(v), 0, rp, ci, v)
    A = P.Matrix(0, m, m, rp, ci, v)
    d = P.Descr(mtype=P.TYPE_TRIANGULAR, fill=P.FILL_UPPER)
    rng = np.random.default_rng(5)
    incb, incx = 2, 3
    b = rng.uniform(-1, 1, m * incb)
    st, xr = oracle.dtrsv("u", 1.5, m, 0, o["val"], o["ind"], o["ptr"], o["iurow"], b[::incb].copy(), False)
    x = np.full(m * incx, 7.0)
    assert P.dtrsv(P.OP_NONE, 1.5, A, d, b, x, incb=incb, incx=incx) == 0
    assert np.array_equal(x[::incx], xr) and np.all(x[1::incx] == 7.0) and np.all(x[2::incx] == 7.0)
    xd = dev(np.full(m * incx, 7.0))
    assert P.dtrsv(P.OP_NONE, 1.5, A, d, dev(b), xd, incb=incb, incx=incx) == 0
    torch.cuda.synchronize()
    assert np.array_equal(xd.cpu().numpy(), x)


def test_block_strsv_float_bit_exact():
    nodes = 3000
    m, rp, ci, v = node_mesh(21, nodes, 45, np.full(nodes, 5))
    vf = v.astype(np.float32)
    o = oracle.dcsr_optimize(m, m, len(v), 0, rp, ci, v)
    A = P.Matrix(0, m, m, rp, ci, vf)
    bf = np.random.default_rng(6).uniform(-1, 1, m).astype(np.float32)
    for fill, fn, iend in ((P.FILL_LOWER, "orc_strsv_l", o["idiag"]), (P.FILL_UPPER, "orc_strsv_u", o["iurow"])):
        for unit in (True, False):
            d = P.Descr(mtype=P.TYPE_TRIANGULAR, fill=fill, diag=P.DIAG_UNIT if unit else P.DIAG_NON_UNIT)
            xo = np.zeros(m, np.float32)
            st = getattr(oracle.lib(), fn)(ctypes.c_float(1.0), m, 0, P._ptr(vf), P._ptr(ci), P._ptr(rp), P._ptr(iend),
                                           P._ptr(bf), 1, P._ptr(xo), 1, 1 if unit else 0)
            assert st == 0
            xd = torch.zeros(m, dtype=torch.float32, device="cuda")
            assert P.strsv(P.OP_NONE, 1.0, A, d, dev(bf), xd) == 0
            torch.cuda.synchronize()
            assert np.array_equal(xd.cpu().numpy(), xo), (fill, unit)


def test_block_trsv_nan_inf_and_tag_propagate():
    """NaN, +-Inf and the exact NOT-READY bit pattern in b: same propagation as the serial chain (a value that is absent
    from a row's chain must not poison it: the kernel multiplies absent entries as 0 * 0, never 0 * x)."""
    nodes = 2000
    m, rp, ci, v = node_mesh(33, nodes, 40, mixed(np.random.default_rng(2), nodes))
    o = oracle.dcsr_optimize(m, m, len(v), 0, rp, ci, v)
    A = P.Matrix(0, m, m, rp, ci, v)
    tag = np.array([0x7FF8DEADBEEF0355], dtype=np.uint64).view(np.float64)[0]
    rng = np.random.default_rng(8)
    for kind, fill, iend in (("l", P.FILL_LOWER, o["idiag"]), ("u", P.FILL_UPPER, o["iurow"])):
        d = P.Descr(mtype=P.TYPE_TRIANGULAR, fill=fill)
        b = rng.uniform(-1, 1, m)
        b[[0, 40, m // 2]] = [np.nan, np.inf, tag]
        b[[m - 1, m - 77]] = [-np.inf, np.nan]
        st, xr = oracle.dtrsv(kind, 1.0, m, 0, o["val"], o["ind"], o["ptr"], iend, b, False)
        xd = torch.zeros(m, dtype=torch.float64, device="cuda")
        assert P.dtrsv(P.OP_NONE, 1.0, A, d, dev(b), xd) == 0
        torch.cuda.synchronize()
        got = xd.cpu().numpy()
        gn, rn = np.isnan(got), np.isnan(xr)
        assert np.array_equal(gn, rn) and np.array_equal(got[~gn], xr[~rn]) and 2 <= rn.sum() < m


@pytest.mark.parametrize("dofs", [(1, 6), (1, 9)])
def test_block_trsv_nan_in_one_component_stays_there(dofs):
    """Two disconnected meshes in one matrix, levels >= 64 blocks wide (every lane of a slice owns a block), NaN / Inf in
    the first mesh only: the second one must come out finite and bit-exact.  (Rows a lane does not own are computed as
    0 - 0 * x like the others and their stores parked behind the solution; a parked NaN landing in the slot that absent
    entries read as 0 would poison every block of the other mesh.)"""
    width, nodes = 160, 160 * 90
    rng = np.random.default_rng(dofs[1])
    m1, rp1, ci1, v1 = node_mesh(91, nodes, width, rng.integers(dofs[0], dofs[1], size=nodes), keep=1.0)
    m = 2 * m1
    rp = np.concatenate([rp1, rp1[1:] + rp1[-1]]).astype(np.int32)
    ci = np.concatenate([ci1, ci1 + m1]).astype(np.int32)
    v = np.concatenate([v1, v1[::-1]])
    rid = np.repeat(np.arange(m), np.diff(rp))
    v[ci == rid] = np.abs(v[ci == rid]) + 2.0
    o = oracle.dcsr_optimize(m, m, len(v), 0, rp, ci, v)
    A = P.Matrix(0, m, m, rp, ci, v)
    for kind, fill, iend, bad in (("l", P.FILL_LOWER, o["idiag"], [0, 3]), ("u", P.FILL_UPPER, o["iurow"], [m1 - 1, m1 - 4])):
        d = P.Descr(mtype=P.TYPE_TRIANGULAR, fill=fill)
        b = rng.uniform(-1, 1, m)
        b[bad] = [np.nan, np.inf]
        st, xr = oracle.dtrsv(kind, 1.0, m, 0, o["val"], o["ind"], o["ptr"], iend, b, False)
        assert st == 0 and np.isfinite(xr[m1:]).all() and np.isnan(xr[:m1]).sum() > m1 // 2
        xd = torch.zeros(m, dtype=torch.float64, device="cuda")
        assert P.dtrsv(P.OP_NONE, 1.0, A, d, dev(b), xd) == 0
        torch.cuda.synchronize()
        got = xd.cpu().numpy()
        assert np.array_equal(got[m1:], xr[m1:]), (kind, int(np.isnan(got[m1:]).sum()))
        gn, rn = np.isnan(got), np.isnan(xr)
        assert np.array_equal(gn, rn) and np.array_equal(got[~gn], xr[~rn])


@pytest.mark.parametrize("order", ["column", "row"])
def test_block_trsm_every_column_bit_exact(order):
    """several right-hand sides: one grid column per right-hand side of the block kernel (own ticket, level counters and
    solution slab); every column equals the serial chain, both dense layouts, padded leading dimensions untouched"""
    nodes = 2500
    m, rp, ci, v = node_mesh(61, nodes, 50, mixed(np.random.default_rng(4), nodes), far=7)
    o = oracle.dcsr_optimize(m, m, len(v), 0, rp, ci, v)
    A = P.Matrix(0, m, m, rp, ci, v)
    n = 7
    rng = np.random.default_rng(12)
    for kind, fill, op, unit in (("l", P.FILL_LOWER, P.OP_NONE, True), ("u", P.FILL_UPPER, P.OP_NONE, False),
                                 ("lt", P.FILL_LOWER, P.OP_TRANSPOSE, False), ("ut", P.FILL_UPPER, P.OP_TRANSPOSE, True)):
        d = P.Descr(mtype=P.TYPE_TRIANGULAR, fill=fill, diag=P.DIAG_UNIT if unit else P.DIAG_NON_UNIT)
        iend = o["idiag"] if kind[0] == "l" else o["iurow"]
        if order == "column":
            ld = m + 3
            Bm = rng.uniform(-1, 1, (n, ld)); Xm = np.full((n, ld), 7.0)
            cols = lambda M, j: M[j, :m]
            pad_ok = lambda M: np.all(M[:, m:] == 7.0)
            lay = P.ORDER_COLUMN
        else:
            ld = n + 2
            Bm = rng.uniform(-1, 1, (m, ld)); Xm = np.full((m, ld), 7.0)
            cols = lambda M, j: M[:, j]
            pad_ok = lambda M: np.all(M[:, n:] == 7.0)
            lay = P.ORDER_ROW
        for ptr_dev in (False, True):
            Bd, Xd = (dev(Bm), dev(Xm)) if ptr_dev else (Bm, Xm.copy())
            assert L.aoclsparse_dtrsm(op, 0.5, A.h, d.h, lay, P._ptr(Bd), n, ld, P._ptr(Xd), ld) == 0
            if ptr_dev:
                torch.cuda.synchronize()
            X = Xd.cpu().numpy() if ptr_dev else Xd
            assert pad_ok(X)
            for j in range(n):
                st, xr = oracle.dtrsv(kind, 0.5, m, 0, o["val"], o["ind"], o["ptr"], iend, np.ascontiguousarray(cols(Bm, j)), unit)
                assert st == 0 and np.array_equal(cols(X, j), xr), (kind, order, ptr_dev, j)
        # a pinned kid: every column in the KT order of that kid (one grid column of trsv_block_kt_kernel per right-hand side)
        Xk = dev(Xm)
        assert L.aoclsparse_dtrsm_kid(op, 0.5, A.h, d.h, lay, P._ptr(dev(Bm)), n, ld, P._ptr(Xk), ld, 3) == 0
        torch.cuda.synchronize()
        X = Xk.cpu().numpy()
        assert pad_ok(X)
        for j in range(n):
            st, xr = oracle.trsv_kt(kind, 8, 0.5, m, 0, o["val"], o["ind"], o["ptr"], iend, np.ascontiguousarray(cols(Bm, j)), unit)
            assert st == 0 and np.array_equal(cols(X, j), xr), (kind, order, "kid 3", j)


_TRACE_SCRIPT = r"""
import os, sys
import numpy as np, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
from test_gpu_trsv_blocks import node_mesh, P, VARIANTS
nodes = 2000
m, rp, ci, v = node_mesh(3, nodes, 40, np.full(nodes, 5))
assert P.lib().aoclsparse_mi355_set_option(P.OPTION_TRSV_CHUNKS, 0) == 0  # (round 6: this test is about the lane-per-block schedule)
A = P.Matrix(0, m, m, rp, ci, v)
b = torch.ones(m, dtype=torch.float64, device="cuda"); x = torch.zeros_like(b)
for kind, fill, op in VARIANTS:
    d = P.Descr(mtype=P.TYPE_TRIANGULAR, fill=getattr(P, fill))
    assert P.dtrsv(getattr(P, op), 1.0, A, d, b, x) == 0
    torch.cuda.synchronize()
    print("solved", kind, flush=True)
"""


def test_block_schedule_is_the_one_that_runs(tmp_path):
    """every (fill, op) triangle of a 5-dof mesh is solved by the block kernel in auto mode: its diagnostic trace is
    written once per solve, and reports blocks of 5 rows"""
    trace = tmp_path / "trace.bin"
    env = dict(os.environ, AOCLSPARSE_MI355_TRSV_TRACE=str(trace))
    r = subprocess.run([sys.executable, "-c", _TRACE_SCRIPT, ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stderr.splitlines() if ln.startswith("[trsv trace]")]
    assert len(lines) == 4 and all("max_rows 5" in ln for ln in lines), r.stderr[-2000:]
    assert trace.stat().st_size > 0 and trace.stat().st_size % 48 == 0


def interleave_two(m, rp, ci, v, seed=1):
    """two independent copies of a matrix with their rows / columns INTERLEAVED (copy c's index i becomes 2 i + c; the second copy
    gets other values): the chained rows of a node -- consecutive in the original -- are two apart now"""
    rng = np.random.default_rng(seed)
    lens = np.diff(rp)
    rp2 = np.zeros(2 * m + 1, np.int64)
    rp2[1:] = np.cumsum(np.repeat(lens, 2))
    ci2 = np.empty(2 * len(ci), np.int64)
    v2 = np.empty(2 * len(ci))
    for i in range(m):
        s, e = rp[i], rp[i + 1]
        for c in range(2):
            d0 = rp2[2 * i + c]
            ci2[d0:d0 + (e - s)] = 2 * ci[s:e].astype(np.int64) + c
            v2[d0:d0 + (e - s)] = v[s:e] if c == 0 else v[s:e] * rng.uniform(0.5, 1.5, e - s)
    return 2 * m, rp2.astype(np.int32), ci2.astype(np.int32), v2


_TRACE_SCRIPT_INTERLEAVED = _TRACE_SCRIPT.replace("from test_gpu_trsv_blocks import node_mesh, P, VARIANTS",
                                                  "from test_gpu_trsv_blocks import node_mesh, interleave_two, P, VARIANTS").replace(
    "A = P.Matrix(0, m, m, rp, ci, v)", "m, rp, ci, v = interleave_two(m, rp, ci, v)\nA = P.Matrix(0, m, m, rp, ci, v)")


def test_block_schedule_finds_chains_that_are_not_numbered_consecutively(tmp_path):
    """Round 4: blocks are chains of the dependency structure (a row continues the block of the row its chain applies last when
    the rest of its list is that row's list), wherever the rows are numbered.  Two interleaved copies of a 5-dof mesh: the dofs
    of a node are two apart, rounds 2-3 (ranges of the solve order) found no block at all; now every triangle runs on the block
    kernel with blocks of 5 rows, bit-exact against the oracle's serial chain."""
    m0, rp0, ci0, v0 = node_mesh(21, 1500, 31, np.full(1500, 5))
    m, rp, ci, v = interleave_two(m0, rp0, ci0, v0)

    def same(got, ref, what):
        assert np.array_equal(got, ref), (what, int((got != ref).sum()))

    solve_all(m, rp, ci, v, same)
    trace = tmp_path / "trace.bin"
    env = dict(os.environ, AOCLSPARSE_MI355_TRSV_TRACE=str(trace))
    r = subprocess.run([sys.executable, "-c", _TRACE_SCRIPT_INTERLEAVED, ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stderr.splitlines() if ln.startswith("[trsv trace]")]
    assert len(lines) == 4 and all("max_rows 5" in ln for ln in lines), r.stderr[-2000:]


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_block_chains_under_random_symmetric_permutations(seed):
    """Any symmetric permutation of a mesh matrix is a valid input whose triangles hold chains in arbitrary places (partly kept,
    partly broken, numbered far apart).  Node-level shuffles (dofs stay together), dof-level shuffles inside windows (chains
    scattered) and full interleaving: every triangle, both diagonal types, automatic and pinned reference kid -- bit-exact
    against the oracle's serial chain whatever blocks the grouping finds."""
    import scipy.sparse as sp

    rng = np.random.default_rng(100 + seed)
    nodes, dofs = 1200, int(rng.integers(2, 7))
    m, rp, ci, v = node_mesh(40 + seed, nodes, int(rng.integers(20, 45)), np.full(nodes, dofs))
    A0 = sp.csr_matrix((v, ci, rp), shape=(m, m))
    if seed == 1:  # nodes shuffled inside windows of 64 nodes, dofs stay consecutive
        pn = np.arange(nodes)
        for w0 in range(0, nodes, 64):
            pn[w0:w0 + 64] = w0 + rng.permutation(min(64, nodes - w0))
        perm = (pn[:, None] * dofs + np.arange(dofs)[None, :]).ravel()
    elif seed == 2:  # rows shuffled inside windows of 4 * dofs rows: the chains of a node are scattered among its neighbours'
        perm = np.arange(m)
        for w0 in range(0, m, 4 * dofs):
            w1 = min(m, w0 + 4 * dofs)
            perm[w0:w1] = w0 + rng.permutation(w1 - w0)
    else:  # dof a of node i -> a * nodes-block interleave of two halves of the mesh
        half = m // 2
        perm = np.concatenate([np.arange(half) * 2, np.arange(m - half) * 2 + 1])
        perm = np.argsort(np.argsort(perm))[:m]
    P_ = sp.csr_matrix((np.ones(m), (perm, np.arange(m))), shape=(m, m))
    Ap = (P_ @ A0 @ P_.T).tocsr()
    Ap.sort_indices()

    def same(got, ref, what):
        assert np.array_equal(got, ref), (seed, what, int((got != ref).sum()))

    solve_all(m, Ap.indptr.astype(np.int32), Ap.indices.astype(np.int32), Ap.data.copy(), same)
