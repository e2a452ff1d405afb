#!/usr/bin/env python3
"""round 4: csr_adaptive_kernel on the 4096^2 Laplacian -- raw aoclsparse_dcsrmv on device arrays vs aoclsparse_dmv on a handle
without a SELL copy (aoclsparse_mi355_set_option(sell, 0)); ms per product between two events, 100 calls back to back."""
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
x = torch.from_numpy(np.sin(0.01 * np.arange(m))).cuda(); y = torch.zeros(m, dtype=torch.float64, device="cuda")
d = pkg.Descr()
def timeit(fn, reps=100):
    for _ in range(10): fn()
    torch.cuda.synchronize(); pkg.timer_start()
    for _ in range(reps): fn()
    return pkg.timer_stop() / reps
out = {"grid": g}
assert L.aoclsparse_mi355_set_option(pkg.OPTION_SELL, 0) == 0
A = pkg.Matrix(0, m, m, rp, ci, v)
assert L.aoclsparse_set_mv_hint(A.h, pkg.OP_NONE, d.h, 100) == 0 and L.aoclsparse_optimize(A.h) == 0
out["handle_dmv_no_sell_ms"] = round(timeit(lambda: pkg.dmv(pkg.OP_NONE, 1.0, A, d, x, 0.0, y)), 5)
inf = A.spmv_info(); out["handle_info"] = {"kernel": inf.kernel, "tile": inf.tile, "row_blocks": inf.row_blocks}
y1 = y.clone()
drp, dci, dv = (torch.from_numpy(a).cuda() for a in (rp, ci, v))
out["raw_dcsrmv_ms"] = round(timeit(lambda: pkg.dcsrmv(pkg.OP_NONE, 1.0, m, m, nnz, dv, dci, drp, d, x, 0.0, y)), 5)
out["same_bits"] = bool(torch.equal(y, y1))
assert L.aoclsparse_mi355_set_option(pkg.OPTION_SELL, -1) == 0
B = pkg.Matrix(0, m, m, rp, ci, v)
assert L.aoclsparse_set_mv_hint(B.h, pkg.OP_NONE, d.h, 100) == 0 and L.aoclsparse_optimize(B.h) == 0
out["handle_dmv_sell_ms"] = round(timeit(lambda: pkg.dmv(pkg.OP_NONE, 1.0, B, d, x, 0.0, y)), 5)
print(json.dumps(out))
