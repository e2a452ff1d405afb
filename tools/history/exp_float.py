#!/usr/bin/env python3
"""Experiment: aoclsparse_smv (fp32) on the headline Laplacian; algorithmic bytes (m+1+nnz)*4 + (m+n+nnz)*4."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package(); L = pkg.lib(); dev = torch.device("cuda", 0)
L.aoclsparse_mi355_set_pointer_mode(pkg.PTR_DEVICE)
d0 = pkg.Descr()
for g in (4096,):
    m, rp, ci, v = entry.laplace5(g, dtype="float32"); nnz = len(v)
    A = pkg.Matrix(0, m, m, rp, ci, v)
    assert L.aoclsparse_set_mv_hint(A.h, pkg.OP_NONE, d0.h, 100) == 0 and L.aoclsparse_optimize(A.h) == 0
    x = torch.rand(m, dtype=torch.float32, device=dev); y = torch.zeros(m, dtype=torch.float32, device=dev)
    for _ in range(5): pkg.smv(pkg.OP_NONE, 1.0, A, d0, x, 0.0, y)
    torch.cuda.synchronize(); pkg.timer_start()
    for _ in range(50): pkg.smv(pkg.OP_NONE, 1.0, A, d0, x, 0.0, y)
    ms = pkg.timer_stop() / 50
    b = (m + 1 + nnz) * 4 + (2 * m + nnz) * 4
    print(json.dumps(dict(op="aoclsparse_smv", grid=g, kernel=A.spmv_info().kernel, order=A.spmv_info().order, ms=round(ms, 4),
                          gflops=round(2.0 * nnz / ms / 1e6, 1), gbs=round(b / ms / 1e6, 1), frac_of_8TBs=round(b / ms / 8e9, 4))))
