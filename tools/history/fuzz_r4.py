#!/usr/bin/env python3
"""round 4: differential fuzz of the kernels added late in the round.
  (a) scsrmm row-major with C read, n >= 256 and a multiple of 4 (four columns per lane) against the two-column kernel on the same
      columns taken 128 at a time: bit for bit; random shapes, leading-dimension paddings, alpha / beta;
  (b) hinted complex ?mv (SELL-64 copy) against the restated operator within (2 len + 16) eps, op N / T / H, random general matrices;
  (c) aoclsparse_dcsr2csc on >= 1 M entries (device sort) against the oracle: bit for bit, random rectangular shapes and bases.
  python3 tools/fuzz_r4.py [iterations=20] [seed=1]"""
import ctypes, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # (tools/history/ -> repository root)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle
from util import pkg, random_csr
P = pkg(); L = P.lib()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
bad = {"scsrmm4": 0, "complex_mv": 0, "csr2csc": 0}
t0 = time.time()
for it in range(iters):
    # (a)
    m, k = int(rng.integers(1, 4000)), int(rng.integers(1, 3000))
    maxlen = int(rng.choice([0, 1, 3, 8, 13, 40, 300]))
    rp, ci, v = random_csr(int(rng.integers(1 << 30)), m, k, lambda r, i: r.integers(0, maxlen + 1))
    A = P.Matrix(0, m, k, rp, ci, v.astype(np.float32)); d = P.Descr()
    n = 256 + 4 * int(rng.integers(0, 80))
    pad = 4 * int(rng.integers(0, 3))
    alpha, beta = (1.0, 0.0) if rng.random() < 0.3 else (float(rng.uniform(-2, 2)), float(rng.uniform(-2, 2)))
    B = rng.uniform(-1, 1, (k, n + pad)).astype(np.float32); C0 = rng.uniform(-1, 1, (m, n + pad)).astype(np.float32)
    Cw = dev(C0.ravel())
    assert P.scsrmm(P.OP_NONE, alpha, A, d, P.ORDER_ROW, dev(B.ravel()), n, n + pad, beta, Cw, n + pad) == 0
    torch.cuda.synchronize()
    W = Cw.cpu().numpy().reshape(m, n + pad)
    ok = np.array_equal(W[:, n:], C0[:, n:])
    for j0 in range(0, n, 128):
        w = min(128, n - j0)
        Bs, Cs0 = np.ascontiguousarray(B[:, j0:j0 + w]), np.ascontiguousarray(C0[:, j0:j0 + w])
        Cs = dev(Cs0.ravel())
        assert P.scsrmm(P.OP_NONE, alpha, A, d, P.ORDER_ROW, dev(Bs.ravel()), w, w, beta, Cs, w) == 0
        torch.cuda.synchronize()
        ok = ok and np.array_equal(Cs.cpu().numpy().reshape(m, w), W[:, j0:j0 + w])
    if not ok:
        bad["scsrmm4"] += 1; print("MISMATCH scsrmm4", it, m, k, maxlen, n, pad, alpha, beta, flush=True)
    # (b)
    prec = "z" if rng.random() < 0.5 else "c"
    dtype, eps = (np.complex128, 2.0 ** -52) if prec == "z" else (np.complex64, float(np.finfo(np.float32).eps))
    create = L.aoclsparse_create_zcsr if prec == "z" else L.aoclsparse_create_ccsr
    mv = L.aoclsparse_zmv if prec == "z" else L.aoclsparse_cmv
    mc, nc = int(rng.integers(1, 900)), int(rng.integers(1, 900))
    rpc, cic, vr = random_csr(int(rng.integers(1 << 30)), mc, nc, lambda r, i: r.integers(0, int(rng.choice([2, 9, 30])) + 1))
    vc = (vr + 1j * rng.uniform(-1, 1, len(vr))).astype(dtype)
    h = ctypes.c_void_p()
    assert create(ctypes.byref(h), 0, mc, nc, len(vc), P._ptr(rpc), P._ptr(cic), P._ptr(vc)) == 0
    al, be = np.array([0.7 - 0.4j], dtype), np.array([-0.3 + 0.2j], dtype)
    lens = np.diff(rpc)
    for opn, op in (("n", P.OP_NONE), ("t", P.OP_TRANSPOSE), ("h", P.OP_CONJ_TRANSPOSE)):
        assert L.aoclsparse_set_mv_hint(h, op, d.h, 10) == 0
        nx, ny = (nc, mc) if opn == "n" else (mc, nc)
        x = (rng.uniform(-1, 1, nx) + 1j * rng.uniform(-1, 1, nx)).astype(dtype)
        y0 = (rng.uniform(-1, 1, ny) + 1j * rng.uniform(-1, 1, ny)).astype(dtype)
        yr, scale = oracle.zmv(opn, "general", "lower", "non_unit", 0, al[0], mc, nc, rpc, cic, vc, x, be[0], y0)
        y = y0.copy()
        assert mv(op, P._ptr(al), h, d.h, P._ptr(x), P._ptr(be), P._ptr(y)) == 0
        colmax = int(np.bincount(cic, minlength=nc).max()) if len(cic) else 0
        bound = (2 * max(int(lens.max()) if len(lens) else 0, colmax, 1) + 16) * eps * (scale + 1e-30)
        if not np.all(np.abs(y - yr) <= bound):
            bad["complex_mv"] += 1; print("MISMATCH complex", it, prec, opn, mc, nc, flush=True)
    L.aoclsparse_destroy(ctypes.byref(h))
# (c) a few large conversions (1-2 M entries each)
for it in range(max(2, iters // 5)):
    m, n = int(rng.integers(60000, 200000)), int(rng.integers(60000, 200000))
    per = (1 << 20) // m + 2 + int(rng.integers(0, 6))
    lens = rng.integers(0, 2 * per, m).astype(np.int64)
    ptr = np.zeros(m + 1, np.int64); np.cumsum(lens, out=ptr[1:])
    nnz = int(ptr[m])
    rows = np.repeat(np.arange(m, dtype=np.int64), lens)
    ind = rng.integers(0, n, nnz)  # unsorted, repeats allowed (off-diagonal repeats are legal; a repeated diagonal only matters to create)
    if rng.random() < 0.5:  # a few hub columns of hundreds of entries
        hub = rng.random(nnz) < 2e-3
        ind[hub] = rng.integers(0, 8, int(hub.sum()))
    val = rng.uniform(-1, 1, nnz)
    bi, bo = int(rng.integers(0, 2)), int(rng.integers(0, 2))
    rp, ci = (ptr + bi).astype(np.int32), (ind + bi).astype(np.int32)
    d = P.Descr(base=bi)
    st, cp, ri, cv = oracle.dcsr2csc(m, n, nnz, bi, bo, rp, ci, val)
    op_, oi, ov = np.zeros(n + 1, np.int32), np.zeros(nnz, np.int32), np.zeros(nnz)
    assert st == 0 and L.aoclsparse_dcsr2csc(m, n, nnz, d.h, bo, P._ptr(rp), P._ptr(ci), P._ptr(val), P._ptr(oi), P._ptr(op_), P._ptr(ov)) == 0
    if not (np.array_equal(op_, cp) and np.array_equal(oi, ri) and np.array_equal(ov, cv)):
        bad["csr2csc"] += 1; print("MISMATCH csr2csc", it, m, n, nnz, bi, bo, flush=True)
print(json.dumps({"tool": "fuzz_r4", "iterations": iters, "mismatches": bad, "seconds": round(time.time() - t0, 1)}))
sys.exit(1 if any(bad.values()) else 0)
