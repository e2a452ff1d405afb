#!/usr/bin/env python3
"""round 4 experiment: the blocked-ELL MFMA csrmm on the block-dense stand-in: python tools/exp_bell.py [cols=256] [nodes_edge=32]
[keep=1.0] [row|col].  (The experiment builds behind profiles/r4/bell_experiments.txt read AOCLSPARSE_MI355_EXP_BELL / _EXP_BELLC once
per process; the shipped library has no such switch.)"""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # (tools/history/ -> repository root)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as entry, standins
pkg = entry.load_package(); L = pkg.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
e = int(sys.argv[2]) if len(sys.argv) > 2 else 32
keep = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
m, rp, ci, v = standins.block_dense(e, e, e, keep=keep)
A = pkg.Matrix(0, m, m, rp, ci, v); d = pkg.Descr()
assert L.aoclsparse_set_mm_hint(A.h, pkg.OP_NONE, d.h, 100) == 0 and L.aoclsparse_optimize(A.h) == 0
L.aoclsparse_mi355_set_pointer_mode(pkg.PTR_DEVICE)
order = pkg.ORDER_COLUMN if len(sys.argv) > 4 and sys.argv[4] == "col" else pkg.ORDER_ROW
ld = m if order == pkg.ORDER_COLUMN else n
B = torch.rand(m * n, dtype=torch.float64, device="cuda") * 2 - 1
C = torch.zeros(m * n, dtype=torch.float64, device="cuda")
res = {}
for ow in (0, 1):
    L.aoclsparse_mi355_set_csrmm_beta0_overwrite(ow)
    for _ in range(3):
        assert pkg.dcsrmm(pkg.OP_NONE, 1.0, A, d, order, B, n, ld, 0.0, C, ld) == 0
    torch.cuda.synchronize(); pkg.timer_start()
    for _ in range(10):
        pkg.dcsrmm(pkg.OP_NONE, 1.0, A, d, order, B, n, ld, 0.0, C, ld)
    ms = pkg.timer_stop() / 10
    res["overwrite" if ow else "c_read"] = {"ms": round(ms, 4), "tflops": round(2.0 * len(v) * n / ms / 1e9, 2)}
print(json.dumps({"exp": os.environ.get("AOCLSPARSE_MI355_EXP_BELLC"), "order": "col" if order == pkg.ORDER_COLUMN else "row", "cols": n, "m": m, "nnz": len(v), "bell_width": A.spmv_info().mm_bell_width, **res}))
