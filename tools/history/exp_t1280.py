#!/usr/bin/env python3
"""round 6 experiment (experiment builds: SPMV_T1280_EXP, SPMV_LINE_EXP=band,blocks per line): raw aoclsparse_dcsrmv on the 4096^2 Laplacian with
1,280-entry tiles on the lines of the grid; checks the bits against the handle path and times cold products."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
pkg = entry.load_package(); L = pkg.lib()
g = 4096
m, rp, ci, v = entry.laplace5(g)
L.aoclsparse_mi355_set_pointer_mode(pkg.PTR_DEVICE)
d = pkg.Descr()
rpd, cid, vd = (torch.from_numpy(a).cuda() for a in (rp, ci, v))
x = torch.rand(m, dtype=torch.float64, device="cuda") * 2 - 1
y = torch.zeros(m, dtype=torch.float64, device="cuda")
A = pkg.Matrix(0, m, m, rp, ci, v)
assert L.aoclsparse_set_mv_hint(A.h, pkg.OP_NONE, d.h, 100) == 0 and L.aoclsparse_optimize(A.h) == 0
yr = torch.zeros(m, dtype=torch.float64, device="cuda")
assert pkg.dmv(pkg.OP_NONE, 1.0, A, d, x, 0.0, yr) == 0
call = lambda: pkg.dcsrmv(pkg.OP_NONE, 1.0, m, m, len(v), vd, cid, rpd, d, x, 0.0, y)
assert call() == 0
torch.cuda.synchronize()
same = bool(torch.equal(y.view(torch.int64), yr.view(torch.int64)))
flush = torch.ones(1 << 28, dtype=torch.float32, device="cuda")
cold = []
for _ in range(12):
    flush.add_(1.0); torch.cuda.synchronize(); pkg.timer_start(); call(); cold.append(pkg.timer_stop())
torch.cuda.synchronize(); pkg.timer_start()
for _ in range(20): call()
b2b = pkg.timer_stop() / 20
print(json.dumps({"t1280": os.environ.get("SPMV_T1280_EXP"), "line": os.environ.get("SPMV_LINE_EXP"), "same_bits": same, "cold_ms": round(float(np.median(cold)), 4), "b2b_ms": round(b2b, 4)}))
