#!/usr/bin/env python3
"""round 4: the other value types on the headline shapes -- complex ?mv (z / c) and float ?csrmm on the g^2 Laplacian, device-resident
operands, ms per call (events, 30 calls after 12) and the fraction of 8 TB/s on CSR-model bytes."""
import ctypes, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # (tools/history/ -> repository root)
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
pkg = entry.load_package(); L = pkg.lib(); P = pkg
g = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
m, rp, ci, v = entry.laplace5(g)
nnz = len(v)
L.aoclsparse_mi355_set_pointer_mode(P.PTR_DEVICE)
out = {"grid": g, "cases": []}
def timed(name, fn, by, reps=30):
    for _ in range(12): assert fn() == 0
    torch.cuda.synchronize(); P.timer_start()
    for _ in range(reps): fn()
    ms = P.timer_stop() / reps
    out["cases"].append({"call": name, "ms": round(ms, 4), "frac_of_8TBs": round(by / (ms * 1e-3) / 8e12, 3)})
d = P.Descr()
for prec, dt, tdt, create, mv in (("z", np.complex128, torch.complex128, L.aoclsparse_create_zcsr, L.aoclsparse_zmv),
                                  ("c", np.complex64, torch.complex64, L.aoclsparse_create_ccsr, L.aoclsparse_cmv)):
    vv = (v * (1 + 0.5j)).astype(dt)
    h = ctypes.c_void_p()
    assert create(ctypes.byref(h), 0, m, m, nnz, P._ptr(rp), P._ptr(ci), P._ptr(vv)) == 0
    assert L.aoclsparse_set_mv_hint(h, P.OP_NONE, d.h, 100) == 0 and L.aoclsparse_optimize(h) == 0
    x = torch.ones(m, dtype=tdt, device="cuda"); y = torch.zeros(m, dtype=tdt, device="cuda")
    a, b = np.array([1.0], dt), np.array([0.0], dt)
    by = (m + 1 + nnz) * 4 + (2 * m + nnz) * vv.itemsize
    timed(prec + "mv N", lambda: mv(P.OP_NONE, P._ptr(a), h, d.h, P._ptr(x), P._ptr(b), P._ptr(y)), by)
    L.aoclsparse_destroy(ctypes.byref(h))
m2, rp2, ci2, v2 = entry.laplace5(1000)
nnz2 = len(v2)
for name, dt, tdt, mm in (("d", np.float64, torch.float64, P.dcsrmm), ("s", np.float32, torch.float32, P.scsrmm)):
    A = P.Matrix(0, m2, m2, rp2, ci2, v2.astype(dt))
    assert L.aoclsparse_set_mm_hint(A.h, P.OP_NONE, d.h, 100) == 0 and L.aoclsparse_optimize(A.h) == 0
    for n in (256, 32):
        B = torch.rand(m2 * n, dtype=tdt, device="cuda"); C = torch.zeros(m2 * n, dtype=tdt, device="cuda")
        es = np.dtype(dt).itemsize
        by = (m2 + 1 + nnz2) * 4 + nnz2 * es + 3 * m2 * n * es
        timed("%scsrmm row-major n=%d" % (name, n), lambda: mm(P.OP_NONE, 1.0, A, d, P.ORDER_ROW, B, n, n, 0.0, C, n), by)
        timed("%scsrmm column-major n=%d" % (name, n), lambda: mm(P.OP_NONE, 1.0, A, d, P.ORDER_COLUMN, B, n, m2, 0.0, C, m2), by)
print(json.dumps(out))
