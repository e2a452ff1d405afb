#!/usr/bin/env python3
"""Experiment: ELLT SpMV vs the CSR-Adaptive dmv on the headline 5-pt Laplacian (device arrays)."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
import oracle
from bench import spmv_bytes
pkg = entry.load_package(); L = pkg.lib(); dev = torch.device("cuda", 0)
L.aoclsparse_mi355_set_pointer_mode(pkg.PTR_DEVICE)
g = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
m, rp, ci, v = entry.laplace5(g); nnz = len(v)
d0 = pkg.Descr()
A = pkg.Matrix(0, m, m, rp, ci, v)
assert L.aoclsparse_set_mv_hint(A.h, pkg.OP_NONE, d0.h, 100) == 0 and L.aoclsparse_optimize(A.h) == 0
xh = np.sin(0.01 * np.arange(m)); x = torch.from_numpy(xh).to(dev); y = torch.zeros(m, dtype=torch.float64, device=dev)
def t(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); pkg.timer_start()
    for _ in range(reps): fn()
    return pkg.timer_stop() / reps
ms = t(lambda: pkg.dmv(pkg.OP_NONE, 1.0, A, d0, x, 0.0, y))
ycsr = y.cpu().numpy().copy()
b = spmv_bytes(m, m, nnz)
print(json.dumps(dict(kind="csr-adaptive", g=g, ms=round(ms, 4), frac=round(b / ms / 1e6 / 8000, 4))))
w = 5
tc = np.empty(m * w, np.int32); tv = np.zeros(m * w)
lens = np.diff(rp)
last = ci[rp[1:] - 1]
for k in range(w):
    has = lens > k
    idx = rp[:-1] + np.minimum(k, lens - 1)
    tc[k * m:(k + 1) * m] = np.where(has, ci[idx], last)
    tv[k * m:(k + 1) * m] = np.where(has, v[idx], 0.0)
tcd, tvd = torch.from_numpy(tc).to(dev), torch.from_numpy(tv).to(dev)
a, bt = np.array([1.0]), np.array([0.0])
y2 = torch.zeros(m, dtype=torch.float64, device=dev)
fn = lambda: L.aoclsparse_delltmv(pkg.OP_NONE, pkg._ptr(a), m, m, nnz, pkg._ptr(tvd), pkg._ptr(tcd), w, d0.h, pkg._ptr(x), pkg._ptr(bt), pkg._ptr(y2))
assert fn() == 0
ms = t(fn)
torch.cuda.synchronize()
print(json.dumps(dict(kind="ellt", g=g, ms=round(ms, 4), frac_csr_bytes=round(b / ms / 1e6 / 8000, 4),
                      ell_bytes=m * w * 12 + 16 * m, frac_ell_bytes=round((m * w * 12 + 16 * m) / ms / 1e6 / 8000, 4),
                      bit_exact_vs_csr=bool(np.array_equal(y2.cpu().numpy(), ycsr)))))
