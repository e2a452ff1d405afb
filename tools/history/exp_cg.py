#!/usr/bin/env python3
"""round 4: aoclsparse_itsol_d_solve (CG, no preconditioner, fixed number of iterations) on the g^2 Laplacian with device vectors:
ms per iteration next to the ms of the SpMV it contains (symmetric descriptor, lower triangle stored)."""
import ctypes, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # (tools/history/ -> repository root)
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
pkg = entry.load_package(); L = pkg.lib(); P = pkg
g = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
n, rp, ci, v = entry.laplace5(g)
rows = np.repeat(np.arange(n), np.diff(rp))
keep = ci <= rows
lrp = np.zeros(n + 1, np.int32); np.cumsum(np.bincount(rows[keep], minlength=n), out=lrp[1:])
A = P.Matrix(0, n, n, lrp, ci[keep].copy(), v[keep].copy())
d = P.Descr(mtype=P.TYPE_SYMMETRIC, fill=P.FILL_LOWER)
L.aoclsparse_mi355_set_pointer_mode(P.PTR_DEVICE)
b = torch.ones(n, dtype=torch.float64, device="cuda")
def solve(limit):
    h = ctypes.c_void_p()
    assert L.aoclsparse_itsol_d_init(ctypes.byref(h)) == 0
    for k, val in {"CG Rel Tolerance": 1e-30, "CG Abs Tolerance": 0.0, "CG Preconditioner": "none", "CG Iteration Limit": limit}.items():
        assert L.aoclsparse_itsol_option_set(h, k.encode(), str(val).encode()) == 0, k
    x = torch.zeros(n, dtype=torch.float64, device="cuda"); rinfo = np.zeros(100)
    torch.cuda.synchronize(); t = time.perf_counter()
    st = L.aoclsparse_itsol_d_solve(h, n, A.h, d.h, P._ptr(b), P._ptr(x), P._ptr(rinfo), None, None, None)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    L.aoclsparse_itsol_destroy(ctypes.byref(h))
    return st, dt, rinfo[30]
solve(5)
st1, t1, it1 = solve(iters)
st2, t2, it2 = solve(2 * iters)
x = torch.ones(n, dtype=torch.float64, device="cuda"); y = torch.zeros(n, dtype=torch.float64, device="cuda")
for _ in range(12): P.dmv(P.OP_NONE, 1.0, A, d, x, 0.0, y)
torch.cuda.synchronize(); P.timer_start()
for _ in range(50): P.dmv(P.OP_NONE, 1.0, A, d, x, 0.0, y)
mv_ms = P.timer_stop() / 50
print(json.dumps({"grid": g, "status": [st1, st2], "iterations": [it1, it2], "ms_per_iteration": round((t2 - t1) * 1e3 / max(it2 - it1, 1), 4),
                  "spmv_ms": round(mv_ms, 4), "vector_bytes_per_iteration_model_MB": round(n * 8 * 10 / 1e6, 1)}))
