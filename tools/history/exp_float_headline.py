#!/usr/bin/env python3
"""round 4: the headline workload in fp32 -- aoclsparse_smv after optimize (SELL-64) and raw aoclsparse_scsrmv (CSR-Adaptive) on
the g^2 5-point Laplacian, next to the fp64 figures of the same process; ms per product between two events, 100 calls back to back.
Bytes: the CSR model of the bench ((m + 1 + nnz) * 4 + (m + n + nnz) * sizeof(T))."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # (tools/history/ -> repository root)
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
pkg = entry.load_package(); L = pkg.lib()
g = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
m, rp, ci, v = entry.laplace5(g)
nnz = len(v)
L.aoclsparse_mi355_set_pointer_mode(pkg.PTR_DEVICE)
d = pkg.Descr()
def timeit(fn, reps=100):
    for _ in range(10): fn()
    torch.cuda.synchronize(); pkg.timer_start()
    for _ in range(reps): fn()
    return pkg.timer_stop() / reps
out = {"grid": g}
for name, dt, mv, csrmv in (("f64", np.float64, pkg.dmv, pkg.dcsrmv), ("f32", np.float32, pkg.smv, pkg.scsrmv)):
    vv = v.astype(dt)
    tdt = torch.float64 if dt == np.float64 else torch.float32
    x = torch.from_numpy(np.sin(0.01 * np.arange(m)).astype(dt)).cuda(); y = torch.zeros(m, dtype=tdt, device="cuda")
    A = pkg.Matrix(0, m, m, rp, ci, vv)
    assert L.aoclsparse_set_mv_hint(A.h, pkg.OP_NONE, d.h, 100) == 0 and L.aoclsparse_optimize(A.h) == 0
    ms = timeit(lambda: mv(pkg.OP_NONE, 1.0, A, d, x, 0.0, y))
    by = (m + 1 + nnz) * 4 + (2 * m + nnz) * vv.itemsize
    inf = A.spmv_info()
    out[name] = {"mv_after_optimize_ms": round(ms, 5), "kernel": inf.kernel, "csr_model_bytes": by, "frac_of_8TBs": round(by / (ms * 1e-3) / 8e12, 4),
                 "gflops": round(2 * nnz / ms / 1e6, 1)}
    drp, dci, dv = (torch.from_numpy(a).cuda() for a in (rp, ci, vv))
    ms2 = timeit(lambda: csrmv(pkg.OP_NONE, 1.0, m, m, nnz, dv, dci, drp, d, x, 0.0, y))
    out[name]["raw_csrmv_ms"] = round(ms2, 5); out[name]["raw_frac_of_8TBs"] = round(by / (ms2 * 1e-3) / 8e12, 4)
    del A
print(json.dumps(out))
