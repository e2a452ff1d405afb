#!/usr/bin/env python3
"""Where does merge-path beat the row-block (CSR-Adaptive) kernel?  Matrices with a few rows far longer than one LDS
tile: the row-block kernel gives such a row to ONE workgroup, merge-path cuts it into 1,024-item tiles spread over the
chip.  One JSON line per (matrix, kernel); the threshold of the automatic choice in csrc/matrix.cpp comes from here."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, [p for p in (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))) if os.path.exists(os.path.join(p, "bench.py"))][0])
import __graft_entry__ as entry
import oracle
from bench import spmv_bytes, timed_laps
pkg = entry.load_package(); L = pkg.lib(); dev = torch.device("cuda", 0)
L.aoclsparse_mi355_set_pointer_mode(pkg.PTR_DEVICE)
d0 = pkg.Descr()


def with_long_rows(n, nlong, length, seed=3):
    """3 entries per row (tridiagonal) + nlong rows of `length` random entries"""
    rng = np.random.default_rng(seed)
    lens = np.full(n, 3, np.int64); lens[0] = lens[-1] = 2
    longs = rng.choice(n, size=nlong, replace=False)
    rows = []
    for i in range(n):
        rows.append(None)
    rp = np.zeros(n + 1, np.int64)
    cols = []
    long_set = set(int(t) for t in longs)
    base = np.arange(n)
    tri = np.stack([base - 1, base, base + 1], axis=1)
    parts = []
    for i in range(n):
        if i in long_set:
            c = np.unique(np.concatenate([rng.integers(0, n, size=length), [i]]))
        else:
            c = tri[i][(tri[i] >= 0) & (tri[i] < n)]
        parts.append(c)
        rp[i + 1] = rp[i] + len(c)
    ci = np.concatenate(parts).astype(np.int32)
    v = rng.uniform(-1, 1, len(ci))
    return n, rp.astype(np.int32), ci, v


cases = [("tridiag + 1 row of 1M", 300000, 1, 1000000), ("tridiag + 4 rows of 250k", 300000, 4, 250000),
         ("tridiag + 16 rows of 64k", 300000, 16, 65536), ("tridiag + 64 rows of 16k", 300000, 64, 16384),
         ("tridiag + 256 rows of 4k", 300000, 256, 4096), ("tridiag + 1024 rows of 1k", 300000, 1024, 1024)]
for title, n, nlong, length in cases:
    m, rp, ci, v = with_long_rows(n, nlong, min(length, n))
    nnz = len(v)
    xh = np.random.default_rng(1).uniform(-1, 1, m)
    x = torch.from_numpy(xh).to(dev); y = torch.zeros(m, dtype=torch.float64, device=dev)
    so, yr = oracle.dcsrmv(0, 0, 1.0, m, nnz, v, ci, rp, xh, 0.0, np.zeros(m))
    for choice in ("adaptive", "merge", "auto"):
        # (round 4: the selection switches of rounds 1-3 are one option hook)
        assert L.aoclsparse_mi355_set_option(pkg.OPTION_SPMV_KERNEL, {"auto": 0, "adaptive": 1, "merge": 2}[choice]) == 0
        assert L.aoclsparse_mi355_set_option(pkg.OPTION_SELL, 0) == 0
        A = pkg.Matrix(0, m, m, rp, ci, v)
        assert L.aoclsparse_set_mv_hint(A.h, pkg.OP_NONE, d0.h, 100) == 0 and L.aoclsparse_optimize(A.h) == 0
        info = A.spmv_info()
        lp = timed_laps(pkg, lambda: pkg.dmv(pkg.OP_NONE, 1.0, A, d0, x, 0.0, y), 50, 5)
        torch.cuda.synchronize()
        err = float(np.max(np.abs(y.cpu().numpy() - yr)))
        print(json.dumps(dict(matrix=title, m=m, nnz=nnz, max_row=int(np.diff(rp).max()), requested=choice,
                              kernel={1: "csr-adaptive", 2: "merge-path", 3: "sell-64"}[info.kernel], tile=info.tile,
                              long_rows=info.long_rows, us=round(float(np.median(lp)) * 1e3, 2),
                              ideal_us=round(spmv_bytes(m, m, nnz) / 6.2e6, 2), max_abs_diff=err)), flush=True)
        del A
