#!/usr/bin/env python3
"""round 6 experiment: row-major csrmm on a mesh stand-in (row groups), MM_GROUP_DEAL_EXP=0 disables the deal of the groups' band to the XCDs
(experiment builds).  python tools/history/exp_mm_standin.py <name> [cols=256]"""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as entry, standins
pkg = entry.load_package(); L = pkg.lib()
name = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
ORD = pkg.ORDER_COLUMN if len(sys.argv) > 3 and sys.argv[3] == "col" else pkg.ORDER_ROW
label, m, rp, ci, v = standins.load(name)
A = pkg.Matrix(0, m, m, rp, ci, v); d = pkg.Descr()
assert L.aoclsparse_set_mm_hint(A.h, pkg.OP_NONE, d.h, 100) == 0 and L.aoclsparse_optimize(A.h) == 0
L.aoclsparse_mi355_set_pointer_mode(pkg.PTR_DEVICE)
B = torch.rand(m * n, dtype=torch.float64, device="cuda") * 2 - 1
C = torch.zeros(m * n, dtype=torch.float64, device="cuda")
LD = m if ORD == pkg.ORDER_COLUMN else n
res = {"layout": "col" if ORD == pkg.ORDER_COLUMN else "row", "matrix": name, "deal": os.environ.get("MM_GROUP_DEAL_EXP", "1"), "cols": n, "groups": int(A.spmv_info().mm_groups)}
for ow in (0, 1):
    L.aoclsparse_mi355_set_csrmm_beta0_overwrite(ow)
    for _ in range(3):
        assert pkg.dcsrmm(pkg.OP_NONE, 1.0, A, d, ORD, B, n, LD, 0.0, C, LD) == 0
    torch.cuda.synchronize(); pkg.timer_start()
    for _ in range(10):
        pkg.dcsrmm(pkg.OP_NONE, 1.0, A, d, ORD, B, n, LD, 0.0, C, LD)
    res["overwrite_ms" if ow else "c_read_ms"] = round(pkg.timer_stop() / 10, 4)
L.aoclsparse_mi355_set_csrmm_beta0_overwrite(0)
res["checksum_bits"] = int(C.view(torch.int64).sum().item())
print(json.dumps(res))
