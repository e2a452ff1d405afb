"""Seeded synthetic CSR generators shared by the CPU and GPU tests (numpy only)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as entry  # noqa: E402

EPS64 = np.finfo(np.float64).eps
EPS32 = np.finfo(np.float32).eps


def pkg():
    return entry.load_package()


laplace5 = entry.laplace5


def random_csr(seed, m, n, rowlen, base=0, dtype=np.float64, sort=True):
    """rowlen: callable(rng, i) -> nnz of row i.  Columns distinct per row."""
    rng = np.random.default_rng(seed)
    lens = np.array([min(n, int(rowlen(rng, i))) for i in range(m)], dtype=np.int64)
    row_ptr = np.zeros(m + 1, dtype=np.int64)
    row_ptr[1:] = np.cumsum(lens)
    col = np.empty(row_ptr[-1], dtype=np.int64)
    for i in range(m):
        k = lens[i]
        if k == 0:
            continue
        if k * 4 < n:
            c = np.unique(rng.integers(0, n, size=k * 2))[:k]
            while len(c) < k:
                c = np.unique(np.concatenate([c, rng.integers(0, n, size=k)]))[:k]
        else:
            c = rng.choice(n, size=k, replace=False)
        c = np.sort(c) if sort else rng.permutation(c)
        col[row_ptr[i]:row_ptr[i + 1]] = c
    val = rng.uniform(-1.0, 1.0, size=row_ptr[-1]).astype(dtype)
    return (row_ptr + base).astype(np.int32), (col + base).astype(np.int32), val


def powerlaw_rows(mean, maxlen):
    def f(rng, i):
        v = int(rng.pareto(1.5) * mean * 0.5) + 1
        return min(v, maxlen)
    return f


def abs_row_sums(row_ptr, col, val, x, base=0):
    """sum_j |a_ij x_j| per row (the scale of the componentwise forward-error bound)."""
    rp = row_ptr.astype(np.int64) - base
    prod = np.abs(val.astype(np.float64) * x[col.astype(np.int64) - base].astype(np.float64))
    out = np.zeros(len(rp) - 1)
    nz = np.diff(rp) > 0
    out[nz] = np.add.reduceat(prod, rp[:-1][nz])
    return out


def triangular_system(seed, m, avg_off, base=0, dtype=np.float64, band=None):
    """Square matrix with a strong diagonal and random off-diagonals on both sides (sorted rows)."""
    rng = np.random.default_rng(seed)
    rows = []
    for i in range(m):
        k = int(rng.integers(0, 2 * avg_off + 1))
        lo = 0 if band is None else max(0, i - band)
        hi = m if band is None else min(m, i + band + 1)
        c = np.unique(np.concatenate([rng.integers(lo, hi, size=k), [i]]))
        rows.append(c)
    lens = np.array([len(c) for c in rows])
    row_ptr = np.zeros(m + 1, dtype=np.int64)
    row_ptr[1:] = np.cumsum(lens)
    col = np.concatenate(rows)
    val = rng.uniform(-0.5, 0.5, size=len(col))
    rid = np.repeat(np.arange(m), lens)
    val[col == rid] = rng.uniform(2.0, 4.0, size=m) * rng.choice([-1.0, 1.0], size=m)
    return (row_ptr + base).astype(np.int32), (col + base).astype(np.int32), val.astype(dtype)


def banded_rows(seed, m, n, per_row, base=0):
    """rows made of short runs of neighbouring columns -- what BLKCSR is meant for"""
    rng = np.random.default_rng(seed)
    rp, ci = [0], []
    for i in range(m):
        cols, want = set(), min(n, int(per_row(rng, i)))
        while len(cols) < want:
            c0 = int(rng.integers(0, n))
            cols.update(range(c0, min(n, c0 + int(rng.integers(1, 7)))))
        cols = sorted(cols)[:want]
        ci += cols
        rp.append(len(ci))
    v = rng.uniform(-1, 1, len(ci))
    return np.array(rp, np.int32) + base, np.array(ci, np.int32) + base, v


class beta0_overwrite:
    """with beta0_overwrite(P): ... -- csrmm with beta == 0 does not read C inside the block (the opt-in mode of
    aoclsparse_mi355_set_csrmm_beta0_overwrite); the default (C read and multiplied by zero, as the reference) is restored."""

    def __init__(self, P):
        self.L = P.lib()

    def __enter__(self):
        assert self.L.aoclsparse_mi355_set_csrmm_beta0_overwrite(1) == 0

    def __exit__(self, *a):
        assert self.L.aoclsparse_mi355_set_csrmm_beta0_overwrite(0) == 0


class trsv_schedule:
    """with trsv_schedule(P, s): ... -- forces TRSV / TRSM schedule s (0 per-level launches, 1 hybrid, 2 sync-free lane per
    position, 3 sync-free slice per wavefront, 4 sync-free lane per block) inside the block; -1 (automatic) is restored."""

    def __init__(self, P, s):
        self.L, self.s = P.lib(), s

    def __enter__(self):
        assert self.L.aoclsparse_mi355_set_trsv_schedule(self.s) == 0

    def __exit__(self, *a):
        assert self.L.aoclsparse_mi355_set_trsv_schedule(-1) == 0


def kt_lanes(kid, dtype=np.float64):
    """vector lanes of the KT kernel the reference dispatches for a pinned kid (trsv.cpp:321-353): None for kid 0 / auto"""
    if kid in (1, 2):
        return 4 if dtype == np.float64 else 8
    if kid == 3:
        return 8 if dtype == np.float64 else 16
    return None
