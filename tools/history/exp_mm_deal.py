#!/usr/bin/env python3
"""round 6 experiment: row-major csrmm on the 1000^2 Laplacian, block order over the XCDs (MM_DEAL_EXP = blocks per XCD turn, experiment
builds only; unset = a contiguous eighth per XCD).  python tools/exp_mm_deal.py [cols=256] [grid=1000]"""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # (tools/history/ -> repository root)
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
pkg = entry.load_package(); L = pkg.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
g = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
m, rp, ci, v = entry.laplace5(g)
A = pkg.Matrix(0, m, m, rp, ci, v); d = pkg.Descr()
assert L.aoclsparse_set_mm_hint(A.h, pkg.OP_NONE, d.h, 100) == 0 and L.aoclsparse_optimize(A.h) == 0
L.aoclsparse_mi355_set_pointer_mode(pkg.PTR_DEVICE)
B = torch.rand(m * n, dtype=torch.float64, device="cuda") * 2 - 1
C = torch.zeros(m * n, dtype=torch.float64, device="cuda")
flush = torch.ones(1 << 28, dtype=torch.float32, device="cuda")
ORD = pkg.ORDER_COLUMN if len(sys.argv) > 3 and sys.argv[3] == "col" else pkg.ORDER_ROW
LD = m if ORD == pkg.ORDER_COLUMN else n
res = {"deal": os.environ.get("MM_DEAL_EXP"), "cols": n, "grid": g}
for ow in (0, 1):
    L.aoclsparse_mi355_set_csrmm_beta0_overwrite(ow)
    for _ in range(3):
        assert pkg.dcsrmm(pkg.OP_NONE, 1.0, A, d, ORD, B, n, LD, 0.0, C, LD) == 0
    torch.cuda.synchronize(); pkg.timer_start()
    for _ in range(20):
        pkg.dcsrmm(pkg.OP_NONE, 1.0, A, d, ORD, B, n, LD, 0.0, C, LD)
    res["overwrite_ms" if ow else "c_read_ms"] = round(pkg.timer_stop() / 20, 4)
    cold = []
    for _ in range(6):
        flush.add_(1.0); torch.cuda.synchronize(); pkg.timer_start()
        pkg.dcsrmm(pkg.OP_NONE, 1.0, A, d, ORD, B, n, LD, 0.0, C, LD)
        cold.append(pkg.timer_stop())
    res["overwrite_cold_ms" if ow else "c_read_cold_ms"] = round(float(np.median(cold)), 4)
res["checksum"] = float(C.sum().item())
print(json.dumps(res))
