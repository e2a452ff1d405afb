#!/usr/bin/env python3
"""round 4: what the one-time calls cost -- aoclsparse_create_dcsr (mat_check), aoclsparse_optimize for an mv / mm / sv hint -- next
to the products they prepare, on the headline Laplacian (grid^2) and the csrmm / TRSV matrices; wall ms, AOCLSPARSE_MI355_TIMING=1
prints the library's own phases."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # (tools/history/ -> repository root)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as entry
pkg = entry.load_package(); L = pkg.lib()
g = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
out = {}
def timed(fn):
    t = time.perf_counter(); r = fn(); return r, round((time.perf_counter() - t) * 1e3, 2)
m, rp, ci, v = entry.laplace5(g)
d = pkg.Descr()
x = np.ones(m); y = np.zeros(m)
for rep in range(2):
    A, t_create = timed(lambda: pkg.Matrix(0, m, m, rp, ci, v))
    _, t_hint = timed(lambda: L.aoclsparse_set_mv_hint(A.h, pkg.OP_NONE, d.h, 100))
    st, t_opt = timed(lambda: L.aoclsparse_optimize(A.h))
    _, t_first = timed(lambda: pkg.dmv(pkg.OP_NONE, 1.0, A, d, x, 0.0, y))
    _, t_second = timed(lambda: pkg.dmv(pkg.OP_NONE, 1.0, A, d, x, 0.0, y))
    out["mv_grid%d_rep%d" % (g, rep)] = {"create_ms": t_create, "optimize_ms": t_opt, "first_dmv_host_vectors_ms": t_first, "second_ms": t_second, "status": st}
    del A
m2, rp2, ci2, v2 = entry.laplace5(1000)
for rep in range(2):
    A, t_create = timed(lambda: pkg.Matrix(0, m2, m2, rp2, ci2, v2))
    L.aoclsparse_set_mm_hint(A.h, pkg.OP_NONE, d.h, 100)
    st, t_opt = timed(lambda: L.aoclsparse_optimize(A.h))
    out["mm_grid1000_rep%d" % rep] = {"create_ms": t_create, "optimize_ms": t_opt, "status": st}
    del A
print(json.dumps(out))
