#!/usr/bin/env python3
"""Experiment: SELL-64 on the shell-like / flan-like stand-ins under each summation order (kid 0 / 1 / 3)."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import __graft_entry__ as entry
import standins
from bench import spmv_bytes
pkg = entry.load_package(); L = pkg.lib(); dev = torch.device("cuda", 0)
L.aoclsparse_mi355_set_pointer_mode(pkg.PTR_DEVICE)
d0 = pkg.Descr()
for name in sys.argv[1:] or ["shell-like"]:
    m, rp, ci, v = standins.ALL[name](); nnz = len(v)
    x = torch.from_numpy(np.random.default_rng(1).uniform(-1, 1, m)).to(dev); y = torch.zeros(m, dtype=torch.float64, device=dev)
    for kid in (0, 1, 3):
        A = pkg.Matrix(0, m, m, rp, ci, v)
        assert L.aoclsparse_set_mv_hint_kid(A.h, pkg.OP_NONE, d0.h, 100, kid) == 0 and L.aoclsparse_optimize(A.h) == 0
        info = A.spmv_info()
        for _ in range(3): pkg.dmv(pkg.OP_NONE, 1.0, A, d0, x, 0.0, y)
        torch.cuda.synchronize(); pkg.timer_start()
        for _ in range(30): pkg.dmv(pkg.OP_NONE, 1.0, A, d0, x, 0.0, y)
        ms = pkg.timer_stop() / 30
        print(json.dumps(dict(matrix=name, kid=kid, kernel=info.kernel, order=info.order, cells_per_nnz=round(info.stored_cells / nnz, 4),
                              ms=round(ms, 4), frac=round(spmv_bytes(m, m, nnz) / ms / 1e6 / 8000, 4))), flush=True)
        del A
