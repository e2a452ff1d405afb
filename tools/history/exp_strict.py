#!/usr/bin/env python3
"""Cost of keeping rows LONGER than one LDS tile in the reference's scalar order (strict long-row path) instead of the
wavefront tree of auto mode, on the two power-law stand-ins.  AOCLSPARSE_MI355_STRICT_LONG=<max row length> (0 = never)."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import __graft_entry__ as entry, oracle, standins
from bench import timed_laps
pkg = entry.load_package(); L = pkg.lib(); dev = torch.device("cuda", 0)
L.aoclsparse_mi355_set_pointer_mode(pkg.PTR_DEVICE)
d0 = pkg.Descr()
for name in ("web-like", "circuit-like"):
    m, rp, ci, v = standins.ALL[name](); nnz = len(v)
    xh = np.random.default_rng(1).uniform(-1, 1, m)
    x = torch.from_numpy(xh).to(dev); y = torch.zeros(m, dtype=torch.float64, device=dev)
    so, yr = oracle.dcsrmv(-1, 0, 1.0, m, nnz, v, ci, rp, xh, 0.0, np.zeros(m), nthreads=8)
    for kid in (-1, 0):
        A = pkg.Matrix(0, m, m, rp, ci, v)
        assert L.aoclsparse_set_mv_hint_kid(A.h, pkg.OP_NONE, d0.h, 100, kid) == 0 and L.aoclsparse_optimize(A.h) == 0
        lp = timed_laps(pkg, lambda: pkg.dmv(pkg.OP_NONE, 1.0, A, d0, x, 0.0, y), 100, 10)
        torch.cuda.synchronize()
        got = y.cpu().numpy()
        print(json.dumps(dict(matrix=name, hinted_kid=kid, env=os.environ.get("AOCLSPARSE_MI355_STRICT_LONG"), us_median=round(float(np.median(lp)) * 1e3, 2),
                              rows_not_bit_exact=int(np.sum(got != yr)), long_rows=int(A.spmv_info().long_rows))), flush=True)
        del A
