#!/usr/bin/env python3
"""round 4: the callers either side of the hot path timed once -- descriptor / operation variants of ?mv, ?dotmv, ?csrmm
with op = T, ?trsv / ?trsm on a triangle, ?symgs -- on the g^2 Laplacian with device-resident vectors; ms per call (events, 20 calls
after 12), first call separately (it holds the one-time work: transposes, derived operators, analyses)."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # (tools/history/ -> repository root)
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
pkg = entry.load_package(); L = pkg.lib(); P = pkg
g = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
m, rp, ci, v = entry.laplace5(g)
nnz = len(v)
L.aoclsparse_mi355_set_pointer_mode(P.PTR_DEVICE)
x = torch.from_numpy(np.sin(0.01 * np.arange(m))).cuda(); y = torch.zeros(m, dtype=torch.float64, device="cuda")
out = {"grid": g, "m": m, "nnz": nnz, "cases": []}
def timed(name, fn, reps=20, bytes_model=None):
    torch.cuda.synchronize(); t = time.perf_counter(); st = fn(); torch.cuda.synchronize(); first = (time.perf_counter() - t) * 1e3
    assert st == 0, (name, st)
    for _ in range(12): fn()  # (past the SELL promotion of an un-hinted handle: 8 products)
    torch.cuda.synchronize(); P.timer_start()
    for _ in range(reps): fn()
    ms = P.timer_stop() / reps
    rec = {"call": name, "first_call_ms": round(first, 2), "ms": round(ms, 4)}
    if bytes_model:
        rec["frac_of_8TBs"] = round(bytes_model / (ms * 1e-3) / 8e12, 3)
    out["cases"].append(rec); return rec
spmv_bytes = (m + 1 + nnz) * 4 + (2 * m + nnz) * 8
A = P.Matrix(0, m, m, rp, ci, v)
dg = P.Descr()
timed("dmv general N (no hint)", lambda: P.dmv(P.OP_NONE, 1.0, A, dg, x, 0.0, y), bytes_model=spmv_bytes)
timed("dmv general T", lambda: P.dmv(P.OP_TRANSPOSE, 1.0, A, dg, x, 0.0, y), bytes_model=spmv_bytes)
dsym = P.Descr(mtype=P.TYPE_SYMMETRIC, fill=P.FILL_LOWER)
timed("dmv symmetric lower N", lambda: P.dmv(P.OP_NONE, 1.0, A, dsym, x, 0.0, y), bytes_model=spmv_bytes)
dtri = P.Descr(mtype=P.TYPE_TRIANGULAR, fill=P.FILL_UPPER)
timed("dmv triangular upper N", lambda: P.dmv(P.OP_NONE, 1.0, A, dtri, x, 0.0, y))
timed("dmv triangular upper T", lambda: P.dmv(P.OP_TRANSPOSE, 1.0, A, dtri, x, 0.0, y))
dot = torch.zeros(1, dtype=torch.float64, device="cuda")
timed("ddotmv general N", lambda: L.aoclsparse_ddotmv(P.OP_NONE, 1.0, A.h, dg.h, P._ptr(x), 0.0, P._ptr(y), P._ptr(dot)), bytes_model=spmv_bytes + 16 * m)
n = 32
B = torch.rand(m * n, dtype=torch.float64, device="cuda"); C = torch.zeros(m * n, dtype=torch.float64, device="cuda")
mm_bytes = (m + 1 + nnz) * 4 + nnz * 8 + 3 * m * n * 8
timed("dcsrmm N row-major n=32", lambda: P.dcsrmm(P.OP_NONE, 1.0, A, dg, P.ORDER_ROW, B, n, n, 0.0, C, n), bytes_model=mm_bytes)
timed("dcsrmm T row-major n=32", lambda: P.dcsrmm(P.OP_TRANSPOSE, 1.0, A, dg, P.ORDER_ROW, B, n, n, 0.0, C, n), bytes_model=mm_bytes)
timed("dcsrmm T column-major n=32", lambda: P.dcsrmm(P.OP_TRANSPOSE, 1.0, A, dg, P.ORDER_COLUMN, B, n, m, 0.0, C, m), bytes_model=mm_bytes)
dlow = P.Descr(mtype=P.TYPE_TRIANGULAR, fill=P.FILL_LOWER)
timed("dtrsv lower N (grid wavefronts: %d levels)" % (2 * g - 1), lambda: P.dtrsv(P.OP_NONE, 1.0, A, dlow, x, y))
timed("dtrsv lower T", lambda: P.dtrsv(P.OP_TRANSPOSE, 1.0, A, dlow, x, y))
nr = 8
Bm = torch.rand(m * nr, dtype=torch.float64, device="cuda"); Xm = torch.zeros(m * nr, dtype=torch.float64, device="cuda")
timed("dtrsm lower N, 8 right-hand sides (column-major)", lambda: L.aoclsparse_dtrsm(P.OP_NONE, 1.0, A.h, dlow.h, P.ORDER_COLUMN, P._ptr(Bm), nr, m, P._ptr(Xm), m))
timed("dsymgs symmetric", lambda: L.aoclsparse_dsymgs(P.OP_NONE, A.h, dsym.h, 1.0, P._ptr(x), P._ptr(y)))
print(json.dumps(out))
