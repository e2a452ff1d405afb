#!/usr/bin/env python3
"""Round-3 differential fuzz (GPU): random shapes through the kernels added this round, against the oracle, bit for bit.
  csrmm_kid row-major (tile / row-per-wave kernels in the csrmm_row_kt arithmetic, tail columns), csrmm default mode with C read
  (row-per-wave C-first kernel), SELL short-row kernel, KT-order TRSV on meshes.   fuzz_r3.py [iterations=40] [seed=1]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # (tools/history/ -> repository root)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle
from util import pkg, random_csr, kt_lanes
from test_gpu_trsv_blocks import node_mesh
P = pkg(); L = P.lib()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for it in range(iters):
    dtype = np.float64 if rng.random() < 0.6 else np.float32
    u = np.uint64 if dtype == np.float64 else np.uint32
    m, k = int(rng.integers(1, 3000)), int(rng.integers(1, 2500))
    maxlen = int(rng.choice([0, 1, 3, 5, 8, 13, 40, 700]))
    rp, ci, v = random_csr(int(rng.integers(1 << 30)), m, k, lambda r, i: r.integers(0, maxlen + 1))
    v = v.astype(dtype)
    A = P.Matrix(0, m, k, rp, ci, v); d = P.Descr()
    n = int(rng.integers(1, 160)) * 2
    alpha, beta = (1.0, 0.0) if rng.random() < 0.3 else (float(rng.uniform(-2, 2)), float(rng.uniform(-2, 2)))
    B = rng.uniform(-1, 1, k * n).astype(dtype); C0 = rng.uniform(-1, 1, m * n).astype(dtype)
    kid = int(rng.integers(1, 4)); lanes = kt_lanes(kid, dtype)
    C = C0.copy()
    fn = P.dcsrmm if dtype == np.float64 else P.scsrmm
    assert fn(P.OP_NONE, alpha, A, d, P.ORDER_ROW, B, n, n, beta, C, n, kid=kid) == 0
    kt = oracle.dcsrmm_kt if dtype == np.float64 else oracle.scsrmm_kt
    st, ref = kt("row", lanes, alpha, 0, v, ci, rp, m, B, n, n, beta, C0, n)
    ok1 = np.array_equal(C.view(u), ref.view(u))
    # default mode (kid None), double only: row-major result equals the column-major reference arithmetic element by element
    ok2 = True
    if dtype == np.float64:
        C = C0.copy()
        assert P.dcsrmm(P.OP_NONE, alpha, A, d, P.ORDER_ROW, B, n, n, beta, C, n) == 0
        Bt = np.ascontiguousarray(B.reshape(k, n).T).ravel(); Ct = np.ascontiguousarray(C0.reshape(m, n).T).ravel()
        st, refc = oracle.dcsrmm("col", alpha, 0, v, ci, rp, m, Bt, n, k, beta, Ct, m)
        ok2 = np.array_equal(np.ascontiguousarray(C.reshape(m, n).T).ravel(), refc)
    if not (ok1 and ok2):
        bad += 1
        print("MISMATCH", it, dtype.__name__, m, k, maxlen, n, kid, alpha, beta, ok1, ok2, flush=True)
print("csrmm fuzz:", iters, "cases,", bad, "mismatches", flush=True)
bad2 = 0
for it in range(max(4, iters // 5)):
    dtype = np.float64 if rng.random() < 0.6 else np.float32
    u = np.uint64 if dtype == np.float64 else np.uint32
    nodes = int(rng.integers(200, 1500)); width = int(rng.integers(5, 60))
    dofs = rng.integers(1, 9, size=nodes) if rng.random() < 0.5 else np.full(nodes, int(rng.integers(1, 9)))
    m, rp, ci, v = node_mesh(int(rng.integers(1 << 30)), nodes, width, dofs, far=int(rng.choice([0, 0, 10, 30])))
    o = oracle.dcsr_optimize(m, m, len(v), 0, rp, ci, v)
    A = P.Matrix(0, m, m, rp, ci, v.astype(dtype))
    for kind, fill in (("l", P.FILL_LOWER), ("u", P.FILL_UPPER)):
        unit = bool(rng.random() < 0.5); kid = int(rng.integers(1, 4))
        d = P.Descr(mtype=P.TYPE_TRIANGULAR, fill=fill, diag=P.DIAG_UNIT if unit else P.DIAG_NON_UNIT)
        b = rng.uniform(-1, 1, m).astype(dtype)
        st, xr = oracle.trsv_kt(kind, kt_lanes(kid, dtype), 0.75, m, 0, o["val"].astype(dtype), o["ind"], o["ptr"],
                                o["idiag"] if kind == "l" else o["iurow"], b, unit, dtype=dtype)
        xd = torch.zeros(m, dtype=torch.float64 if dtype == np.float64 else torch.float32, device="cuda")
        solve = P.dtrsv if dtype == np.float64 else P.strsv
        assert solve(P.OP_NONE, 0.75, A, d, torch.from_numpy(b).cuda(), xd, kid=kid) == 0
        torch.cuda.synchronize()
        if not np.array_equal(xd.cpu().numpy().view(u), xr.view(u)):
            bad2 += 1
            print("TRSV MISMATCH", it, dtype.__name__, m, kind, unit, kid, flush=True)
print("trsv fuzz:", bad2, "mismatches", flush=True)
sys.exit(1 if bad or bad2 else 0)
