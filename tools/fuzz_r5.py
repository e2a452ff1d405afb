#!/usr/bin/env python3
"""round 5: differential fuzz of the kernels changed this round, against the CPU oracle.
  (a) aoclsparse_dmv without a kernel id on random power-law matrices (both bases, alpha / beta classes, empty rows, rows of 1 .. 20,000
      entries): the automatic kernel (CSR-Adaptive with the wavefront tree, or ONE-launch merge-path when the longest row spans >= 16
      tiles) and merge-path forced -- rows of fewer than 32 entries that no merge tile cuts: bit for bit; every row within
      (2 ceil(log2 n) + 4 + n / 256 + pieces + 2) eps sum|a x|; with spmv_strict every row bit for bit;
  (b) the blocked-ELL MFMA csrmm on random block-dense matrices (random node grids, tile fill 0.55 .. 1, both layouts, column counts
      that are and are not multiples of 16, padded leading dimensions, the three beta modes) with Inf / NaN scattered over B and C:
      finite exactly where the reference is finite, bit for bit there, NaN where it has NaN;
  (c) aoclsparse_dcsr2csc on >= 1 M entries with rows of thousands of entries (the wavefront-per-long-row path of the device sort).
  python3 tools/fuzz_r5.py [iterations=20] [seed=1]"""
import ctypes, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import oracle, standins
from util import pkg, random_csr
P = pkg(); L = P.lib()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
EPS = 2.0 ** -52
bad = {"spmv_auto": 0, "spmv_merge": 0, "spmv_strict": 0, "bell": 0, "csr2csc": 0}
ran = {"spmv": 0, "spmv_merge_kernel": 0, "tree_rows": 0, "bell": 0, "bell_nonfinite_results": 0, "csr2csc": 0}
t0 = time.time()


def merge_cut_rows(rp, base, items=1024):
    m = len(rp) - 1
    end = rp[1:].astype(np.int64) - base
    key = end + np.arange(m)
    d = np.arange(0, m + end[-1] + items, items)
    d = d[d <= m + end[-1]]
    i = np.searchsorted(key, d, side="left")
    j = d - i
    ok = i < m
    cut = np.zeros(m, dtype=bool)
    ii = i[ok]
    cut[ii[(rp[ii].astype(np.int64) - base) < j[ok]]] = True
    return cut


for it in range(iters):
    # ---- (a) --------------------------------------------------------------------------------------------------------------
    m = int(rng.integers(2000, 60000)); n = int(rng.integers(max(2000, m // 2), 2 * m))
    base = int(rng.integers(0, 2))
    longest = int(rng.choice([40, 300, 1500, 9000, 20000]))
    nlong = int(rng.integers(1, 6))
    long_rows = set(int(t) for t in rng.choice(m, size=nlong, replace=False))
    mean = float(rng.choice([2.0, 5.0, 9.0]))

    def lens_of(r, i, longest=longest, long_rows=long_rows, mean=mean):
        if i in long_rows:
            return min(longest, n)
        if r.random() < 0.1:
            return 0
        return int(min(n, max(1, r.pareto(1.6) * mean * 0.4 + 1)))
    rp, ci, v = random_csr(int(rng.integers(1 << 30)), m, n, lens_of, base=base)
    if len(v) > 10 * m:
        continue  # the reference would dispatch the 8-lane order: not what this fuzz is about
    nnz = len(v)
    x = rng.uniform(-1, 1, n); y0 = rng.uniform(-1, 1, m)
    alpha, beta = (1.0, 0.0) if rng.random() < 0.4 else (float(rng.uniform(-2, 2)), float(rng.uniform(-2, 2)))
    so, yr = oracle.dcsrmv(-1, base, alpha, m, nnz, v, ci, rp, x, beta, y0)
    lens = np.diff(rp).astype(np.int64)
    scale = np.zeros(m); nzr = lens > 0
    scale[nzr] = np.add.reduceat(np.abs(v * x[ci - base]), (rp[:-1] - base)[nzr])
    cut = merge_cut_rows(rp, base)
    bound = (2 * np.ceil(np.log2(np.maximum(lens, 2))) + 6 + lens / 256.0 + lens / 1024.0) * EPS * abs(alpha) * scale + 2 * EPS * np.abs(beta * y0)
    d = P.Descr(base=base)
    for mode in ("auto", "merge", "strict"):
        assert L.aoclsparse_mi355_set_option(P.OPTION_SPMV_KERNEL, 2 if mode == "merge" else 0) == 0
        assert L.aoclsparse_mi355_set_option(P.OPTION_SELL, 0) == 0
        assert L.aoclsparse_mi355_set_option(P.OPTION_SPMV_STRICT, 1 if mode == "strict" else 0) == 0
        try:
            A = P.Matrix(base, m, n, rp, ci, v)
            assert L.aoclsparse_set_mv_hint(A.h, P.OP_NONE, d.h, 10) == 0 and L.aoclsparse_optimize(A.h) == 0
            yd = dev(y0)
            assert P.dmv(P.OP_NONE, alpha, A, d, dev(x), beta, yd) == 0
            torch.cuda.synchronize()
            y = yd.cpu().numpy()
            kern = A.spmv_info().kernel
        finally:
            L.aoclsparse_mi355_set_option(P.OPTION_SPMV_KERNEL, 0); L.aoclsparse_mi355_set_option(P.OPTION_SELL, -1)
            L.aoclsparse_mi355_set_option(P.OPTION_SPMV_STRICT, 0)
        ran["spmv"] += 1
        ran["spmv_merge_kernel"] += kern == 2
        ran["tree_rows"] += int((lens >= 32).sum()) if mode == "auto" else 0
        if mode == "strict":
            ok = np.array_equal(y, yr)
        else:
            exact = (lens < 32) & (~cut if kern == 2 else np.ones(m, bool))
            ok = np.array_equal(y[exact], yr[exact]) and bool(np.all(np.abs(y - yr) <= bound + 1e-300))
        if not ok:
            bad["spmv_" + mode] += 1
            print("MISMATCH spmv", mode, it, m, n, base, longest, nlong, alpha, beta, kern, flush=True)
    # ---- (b) --------------------------------------------------------------------------------------------------------------
    nx, ny, nz = (int(t) for t in rng.integers(3, 8, 3))
    keep = float(rng.choice([1.0, 0.9, 0.75, 0.6]))
    mb, rpb, cib, vb = standins.block_dense(nx, ny, nz, keep=keep, seed=int(rng.integers(1 << 20)))
    Ab = P.Matrix(0, mb, mb, rpb, cib, vb); d0 = P.Descr()
    assert L.aoclsparse_set_mm_hint(Ab.h, P.OP_NONE, d0.h, 10) == 0 and L.aoclsparse_optimize(Ab.h) == 0
    if mb >= 1024 and Ab.spmv_info().mm_bell_width > 0:
        nc = int(rng.choice([7, 16, 30, 32, 48, 64, 70, 128]))
        colmaj = rng.random() < 0.5
        B = rng.uniform(-1, 1, (mb, nc)); C0 = rng.uniform(-1, 1, (mb, nc))
        for arr, cnt in ((B, int(rng.integers(0, 30))), (C0, int(rng.integers(0, 6)))):
            for _ in range(cnt):
                arr[int(rng.integers(0, mb)), int(rng.integers(0, nc))] = rng.choice([np.inf, -np.inf, np.nan])
        mode = int(rng.integers(0, 3))  # 0: beta = 0 C read, 1: beta = 0 overwritten, 2: beta != 0
        alpha = float(rng.choice([1.0, -0.5, 2.0])); beta = 0.0 if mode < 2 else float(rng.choice([1.0, -1.5]))
        Bc, Cc = np.ascontiguousarray(B.T).ravel(), np.ascontiguousarray(C0.T).ravel()
        Cin = np.where(np.isfinite(Cc), Cc, 0.25) if mode == 1 else Cc
        so, Cr = oracle.dcsrmm("col", alpha, 0, vb, cib, rpb, mb, Bc, nc, mb, beta, Cin, mb)
        ref = Cr.reshape(nc, mb).T
        assert L.aoclsparse_mi355_set_csrmm_beta0_overwrite(1 if mode == 1 else 0) == 0
        try:
            if colmaj:
                pad = int(rng.integers(0, 3))
                Bp = np.zeros((nc, mb + pad)); Bp[:, :mb] = B.T
                Cp = np.full((nc, mb + pad), 7.0); Cp[:, :mb] = C0.T
                Cd = dev(Cp.ravel())
                assert P.dcsrmm(P.OP_NONE, alpha, Ab, d0, P.ORDER_COLUMN, dev(Bp.ravel()), nc, mb + pad, beta, Cd, mb + pad) == 0
                torch.cuda.synchronize()
                got = Cd.cpu().numpy().reshape(nc, mb + pad)[:, :mb].T
            else:
                pad = int(rng.integers(0, 3))
                Bp = np.zeros((mb, nc + pad)); Bp[:, :nc] = B
                Cp = np.full((mb, nc + pad), 7.0); Cp[:, :nc] = C0
                Cd = dev(Cp.ravel())
                assert P.dcsrmm(P.OP_NONE, alpha, Ab, d0, P.ORDER_ROW, dev(Bp.ravel()), nc, nc + pad, beta, Cd, nc + pad) == 0
                torch.cuda.synchronize()
                full = Cd.cpu().numpy().reshape(mb, nc + pad)
                got = full[:, :nc]
                if not np.all(full[:, nc:] == 7.0):
                    bad["bell"] += 1; print("PADDING touched", it, flush=True)
        finally:
            L.aoclsparse_mi355_set_csrmm_beta0_overwrite(0)
        fin = np.isfinite(ref)
        ran["bell"] += 1
        ran["bell_nonfinite_results"] += int((~fin).sum())
        if not (np.array_equal(np.isfinite(got), fin) and np.array_equal(got[fin], ref[fin]) and np.array_equal(np.isnan(got), np.isnan(ref))
                and np.array_equal(got[np.isinf(ref)], ref[np.isinf(ref)])):
            bad["bell"] += 1; print("MISMATCH bell", it, nx, ny, nz, keep, nc, "col" if colmaj else "row", mode, alpha, beta, flush=True)
# ---- (c) a few large conversions whose rows are long --------------------------------------------------------------------------
for it in range(max(2, iters // 5)):
    m, n = int(rng.integers(3000, 9000)), int(rng.integers(150000, 400000))
    per = (1 << 20) // m + 50
    lens = rng.integers(per // 2, 3 * per // 2, m).astype(np.int64)
    lens[rng.integers(0, m, 5)] = rng.integers(2000, 20000, 5)  # a handful of very long rows
    ptr = np.zeros(m + 1, np.int64); np.cumsum(lens, out=ptr[1:])
    nnz = int(ptr[m])
    ind = rng.integers(0, n, nnz)
    val = rng.uniform(-1, 1, nnz)
    bi, bo = int(rng.integers(0, 2)), int(rng.integers(0, 2))
    rp, ci = (ptr + bi).astype(np.int32), (ind + bi).astype(np.int32)
    dd = P.Descr(base=bi)
    st, cp, ri, cv = oracle.dcsr2csc(m, n, nnz, bi, bo, rp, ci, val)
    op_, oi, ov = np.zeros(n + 1, np.int32), np.zeros(nnz, np.int32), np.zeros(nnz)
    assert st == 0 and L.aoclsparse_dcsr2csc(m, n, nnz, dd.h, bo, P._ptr(rp), P._ptr(ci), P._ptr(val), P._ptr(oi), P._ptr(op_), P._ptr(ov)) == 0
    ran["csr2csc"] += 1
    if not (np.array_equal(op_, cp) and np.array_equal(oi, ri) and np.array_equal(ov, cv)):
        bad["csr2csc"] += 1; print("MISMATCH csr2csc", it, m, n, nnz, bi, bo, flush=True)
print(json.dumps({"tool": "fuzz_r5", "iterations": iters, "checked": {k: int(v) for k, v in ran.items()}, "mismatches": bad, "seconds": round(time.time() - t0, 1)}))
sys.exit(1 if any(bad.values()) else 0)
