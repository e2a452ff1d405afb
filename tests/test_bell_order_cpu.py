"""The order of the blocked-ELL block rows over the XCDs (csrmm_api.cpp choose_bell_order) through its host-only entry point
aoclsparse_mi355_plan_block_row_order: no device involved, so the model, the lattice detection and the lists are checked here, on the
CPU tier (and under the sanitizer build)."""
import ctypes

import numpy as np
import pytest

from util import pkg

P = pkg()
L = P.lib()
AUTO = -2


def grid_bcol(nx, ny, nz, stencil=7, drop=None, seed=0):
    """block columns of a stencil on an nx x ny x nz grid numbered x fastest (block row = x + nx (y + ny z)); ascending, -1 padded"""
    if stencil == 7:
        offs = [(0, 0, -1), (0, -1, 0), (-1, 0, 0), (0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1)]
    elif stencil == 5:
        offs = [(0, -1, 0), (-1, 0, 0), (0, 0, 0), (1, 0, 0), (0, 1, 0)]
    else:
        offs = [(dx, dy, dz) for dz in (-1, 0, 1) for dy in (-1, 0, 1) for dx in (-1, 0, 1)]
    n = nx * ny * nz
    idx = np.arange(n)
    x, y, z = idx % nx, (idx // nx) % ny, idx // (nx * ny)
    cols = np.full((n, len(offs)), -1, dtype=np.int64)
    for k, (dx, dy, dz) in enumerate(offs):
        ok = (x + dx >= 0) & (x + dx < nx) & (y + dy >= 0) & (y + dy < ny) & (z + dz >= 0) & (z + dz < nz)
        cols[ok, k] = (x + dx + nx * (y + dy + ny * (z + dz)))[ok]
    if drop:
        rng = np.random.default_rng(seed)
        kill = (rng.random(cols.shape) < drop) & (cols != idx[:, None])
        cols[kill] = -1
    big = np.where(cols < 0, np.iinfo(np.int64).max, cols)
    big.sort(axis=1)
    out = np.where(big == np.iinfo(np.int64).max, -1, big).astype(np.int32)
    return n, out.shape[1], np.ascontiguousarray(out)


def plan(nbr, width, nbc, bcol, forced=AUTO, capacity=None):
    cap = 8 * nbr if capacity is None else capacity
    order = np.full(max(cap, 1), -7, dtype=np.int32)
    olen = ctypes.c_int32(-1)
    info = (ctypes.c_int32 * 8)()
    st = L.aoclsparse_mi355_plan_block_row_order(nbr, width, nbc, bcol.ctypes.data, forced, order.ctypes.data, cap, ctypes.byref(olen), info)
    return st, order, olen.value, list(info)


def check_lists(nbr, order, olen):
    """every block row exactly once; XCD x's list is order[x::8][:olen] with -1 only at its end; no XCD above 1.06 of its share"""
    lists = order[: 8 * olen].reshape(olen, 8).T
    seen = np.zeros(nbr, dtype=np.int64)
    longest = 0
    for l in lists:
        k = int((l >= 0).sum())
        assert (l[:k] >= 0).all() and (l[k:] == -1).all()
        np.add.at(seen, l[:k], 1)
        longest = max(longest, k)
    assert (seen == 1).all()
    assert longest <= 1.06 * nbr / 8 + 1
    return lists


def test_seven_point_grid_gets_the_lattice_sweep():
    nbr, width, bcol = grid_bcol(32, 32, 32)
    st, order, olen, info = plan(nbr, width, nbr, bcol)
    assert st == 0 and info[0] == 0 and info[1:4] == [32, 32, 32]
    assert info[4] == 4 and 4 <= info[5] <= 10  # 8 pieces per line of 32, about 40 block rows per region
    assert 4500 <= info[7] <= 5200 and info[6] < 2000  # model: ~4.9 fetches per B block row in launch order, under 2 in the sweep
    lists = check_lists(nbr, order, olen)
    # XCD 0 follows its first region through the planes: positions a*b apart are one plane apart
    a, b = info[4], info[5]
    assert lists[0][a * b] - lists[0][0] == 32 * 32
    # launch order asked for: no list
    st, _, olen0, info0 = plan(nbr, width, nbr, bcol, forced=0)
    assert st == 0 and olen0 == 0 and info0[0] == 1 and info0[6] == info0[7]


@pytest.mark.parametrize("dims,stencil,drop", [((40, 40, 40), 7, None), ((24, 24, 24), 27, None), ((20, 24, 16), 7, 0.25), ((27, 16, 8), 7, None)])
def test_lattices_are_recognised_by_their_offsets(dims, stencil, drop):
    nbr, width, bcol = grid_bcol(*dims, stencil=stencil, drop=drop, seed=4)
    st, order, olen, info = plan(nbr, width, nbr, bcol, forced=-1)
    assert st == 0 and info[0] == 0 and info[1:4] == [dims[0], dims[1], dims[2]], info
    check_lists(nbr, order, olen)


def test_two_dimensional_grid_and_grids_too_small_to_balance():
    nbr, width, bcol = grid_bcol(64, 64, 1, stencil=5)
    st, order, olen, info = plan(nbr, width, nbr, bcol, forced=-1)
    assert st == 0 and info[0] == 0 and info[1:4] == [64, 1, 64]
    check_lists(nbr, order, olen)
    # 9 x 14 x 13: the regions do not go round the eight XCDs evenly -> no sweep even when asked for; whatever is picked is a permutation
    nbr, width, bcol = grid_bcol(13, 14, 9)
    st, order, olen, info = plan(nbr, width, nbr, bcol, forced=-1)
    assert st == 0 and info[1] == 0 and info[0] >= 1
    if olen:
        check_lists(nbr, order, olen)
    # under 512 block rows: launch order, nothing modelled
    nbr, width, bcol = grid_bcol(6, 6, 6)
    st, _, olen, info = plan(nbr, width, nbr, bcol)
    assert st == 0 and olen == 0 and info[0] == 1


def test_chunks_for_matrices_without_a_lattice_and_forced_chunks():
    rng = np.random.default_rng(2)
    nbr, width = 4000, 6
    # a band of random neighbours: block row k stores k and five of k - 40 .. k + 40
    cols = np.empty((nbr, width), dtype=np.int64)
    for k in range(nbr):
        nb = np.unique(np.clip(k + rng.integers(-40, 41, size=width - 1), 0, nbr - 1))
        nb = np.union1d(nb, [k])[:width]
        cols[k] = np.concatenate([nb, np.full(width - len(nb), np.iinfo(np.int64).max)])
    bcol = np.where(cols == np.iinfo(np.int64).max, -1, cols).astype(np.int32)
    st, order, olen, info = plan(nbr, width, nbr, bcol)
    assert st == 0 and info[1] == 0 and info[0] >= 1 and info[6] <= info[7]
    if olen:
        check_lists(nbr, order, olen)
    for forced in (3, 7, 64):  # chunks that do not divide 4,000 block rows
        st, order, olen, info = plan(nbr, width, nbr, bcol, forced=forced)
        assert st == 0 and info[0] == forced and olen > 0
        lists = order[: 8 * olen].reshape(olen, 8).T
        seen = np.zeros(nbr, dtype=np.int64)
        for l in lists:
            np.add.at(seen, l[l >= 0], 1)
        assert (seen == 1).all()
        assert list(lists[1][:forced]) == list(range(forced, 2 * forced))  # XCD 1 starts with the second chunk


def test_plan_block_row_order_checks_its_arguments():
    nbr, width, bcol = grid_bcol(16, 16, 16)
    assert plan(nbr, width, nbr, bcol, forced=2, capacity=8)[0] != 0  # a list of 8 * 256 entries does not fit 8
    bad = bcol.copy()
    bad[5, 0] = nbr + 3
    assert plan(nbr, width, nbr, bad)[0] != 0
    olen = ctypes.c_int32()
    info = (ctypes.c_int32 * 8)()
    assert L.aoclsparse_mi355_plan_block_row_order(nbr, width, nbr, None, AUTO, None, 0, ctypes.byref(olen), info) != 0
    assert L.aoclsparse_mi355_plan_block_row_order(nbr, 0, nbr, bcol.ctypes.data, AUTO, None, 0, ctypes.byref(olen), info) != 0
