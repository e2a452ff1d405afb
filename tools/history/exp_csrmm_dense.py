#!/usr/bin/env python3
"""Experiment: row-major csrmm (256 columns) on matrices with more non-zeros per row than the 5-pt Laplacian."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import __graft_entry__ as entry
import standins
from bench import csrmm_bytes
pkg = entry.load_package(); L = pkg.lib(); dev = torch.device("cuda", 0)
L.aoclsparse_mi355_set_pointer_mode(pkg.PTR_DEVICE)
d0 = pkg.Descr()
n = int(os.environ.get("NCOLS", "256"))
for name in sys.argv[1:] or ["shell-like", "flan-like"]:
    m, rp, ci, v = standins.ALL[name](); nnz = len(v)
    A = pkg.Matrix(0, m, m, rp, ci, v)
    assert L.aoclsparse_set_mm_hint(A.h, pkg.OP_NONE, d0.h, 10) == 0 and L.aoclsparse_optimize(A.h) == 0
    gen = torch.Generator(device=dev); gen.manual_seed(7)
    B = torch.rand(m * n, dtype=torch.float64, device=dev, generator=gen) * 2 - 1
    C = torch.zeros(m * n, dtype=torch.float64, device=dev)
    for order, ld, nm in ((pkg.ORDER_ROW, n, "row-major"), (pkg.ORDER_COLUMN, m, "column-major")):
        for _ in range(2): pkg.dcsrmm(pkg.OP_NONE, 1.0, A, d0, order, B, n, ld, 0.0, C, ld)
        torch.cuda.synchronize(); pkg.timer_start()
        for _ in range(5): pkg.dcsrmm(pkg.OP_NONE, 1.0, A, d0, order, B, n, ld, 0.0, C, ld)
        ms = pkg.timer_stop() / 5
        b = csrmm_bytes(m, m, nnz, n, False)
        print(json.dumps(dict(matrix=name, m=m, nnz=nnz, layout=nm, ms=round(ms, 3), gflops=round(2.0 * nnz * n / ms / 1e6, 1),
                              gbs=round(b / ms / 1e6, 1), frac_of_8TBs=round(b / ms / 8e9, 4), hbm_floor_ms=round(b / 6.2e9, 3))), flush=True)
    del A, B, C
