#!/usr/bin/env python3
"""Experiment: row-major csrmm on 256 columns as 1/2/4/8 column-block passes (strided views of B and C)."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
from bench import csrmm_bytes
pkg = entry.load_package(); L = pkg.lib(); dev = torch.device("cuda", 0)
L.aoclsparse_mi355_set_pointer_mode(pkg.PTR_DEVICE)
g = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
m, rp, ci, v = entry.laplace5(g); nnz = len(v); n = 256
A = pkg.Matrix(0, m, m, rp, ci, v); d0 = pkg.Descr()
assert L.aoclsparse_set_mm_hint(A.h, pkg.OP_NONE, d0.h, 10) == 0 and L.aoclsparse_optimize(A.h) == 0
B = torch.rand(m * n, dtype=torch.float64, device=dev) * 2 - 1
C = torch.zeros(m * n, dtype=torch.float64, device=dev)
Cref = None
for beta in (0.0, -2.0):
  for nb in (1, 2, 4, 8):
    w = n // nb
    def run():
        for j in range(nb):
            pkg.dcsrmm(pkg.OP_NONE, 1.0, A, d0, pkg.ORDER_ROW, B[j * w:], w, n, beta, C[j * w:], n)
    C.zero_()
    for _ in range(2): run()
    torch.cuda.synchronize(); pkg.timer_start()
    for _ in range(10): run()
    ms = pkg.timer_stop() / 10
    b = csrmm_bytes(m, m, nnz, n, beta != 0.0)
    print(json.dumps(dict(beta=beta, blocks=nb, width=w, ms=round(ms, 4), frac=round(b / ms / 1e6 / 8000, 4))), flush=True)
