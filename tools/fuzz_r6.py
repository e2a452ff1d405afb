#!/usr/bin/env python3
"""Round-6 GPU fuzzer for the csrmm order plans: random banded matrices (5- and 9-point stencils on gx x gy grids, some rows thinned, optional
node blocks of several unknowns) x random column counts / layouts / beta classes / both beta = 0 modes, every product checked bit for bit
against oracle.dcsrmm on a few columns.  python tools/fuzz_r6.py [cases=24] [seed=1]"""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as entry
import oracle
P = entry.load_package(); L = P.lib()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)


def stencil(gx, gy, nine, dofs, thin):
    nodes = gx * gy
    i = np.arange(nodes)
    x, y = i % gx, i // gx
    offs = [(dx, dy) for dy in (-1, 0, 1) for dx in (-1, 0, 1) if nine or dx == 0 or dy == 0]
    rows, cols = [], []
    for dx, dy in offs:
        ok = (x + dx >= 0) & (x + dx < gx) & (y + dy >= 0) & (y + dy < gy)
        if thin and (dx, dy) != (0, 0):
            ok &= rng.random(nodes) > thin
        rows.append(i[ok]); cols.append((i + dx + gx * dy)[ok])
    r, c = np.concatenate(rows), np.concatenate(cols)
    if dofs > 1:  # dense dofs x dofs node blocks
        a, b = np.meshgrid(np.arange(dofs), np.arange(dofs), indexing="ij")
        r = (r[:, None, None] * dofs + a[None]).ravel(); c = (c[:, None, None] * dofs + b[None]).ravel()
    m = nodes * dofs
    key = r * m + c
    key.sort()
    r, c = key // m, key % m
    rp = np.zeros(m + 1, dtype=np.int64); np.add.at(rp, r + 1, 1)
    return m, np.cumsum(rp).astype(np.int32), c.astype(np.int32), rng.uniform(-1, 1, len(c))


L.aoclsparse_mi355_set_pointer_mode(P.PTR_DEVICE)
bad = 0
for case in range(cases):
    gx = int(rng.integers(260, 700)); gy = int(rng.integers(40, 200))
    nine = bool(rng.integers(0, 2)); dofs = int(rng.choice([1, 1, 2, 3])); thin = float(rng.choice([0.0, 0.0, 0.05]))
    if dofs > 1:
        gx, gy = max(260 // dofs + 20, gx // 3), max(30, gy // 2)
    m, rp, ci, v = stencil(gx, gy, nine, dofs, thin)
    A = P.Matrix(0, m, m, rp, ci, v); d = P.Descr()
    assert L.aoclsparse_set_mm_hint(A.h, P.OP_NONE, d.h, 10) == 0 and L.aoclsparse_optimize(A.h) == 0
    for n in rng.choice([32, 40, 48, 64, 96, 128, 130, 192, 256], size=3, replace=False):
        n = int(n)
        colmaj = bool(rng.integers(0, 4) == 0)
        order, ld = (P.ORDER_COLUMN, m) if colmaj else (P.ORDER_ROW, n)
        alpha, beta = float(rng.choice([1.0, -0.5, 2.0])), float(rng.choice([0.0, 0.0, 1.25]))
        B = torch.rand(m * n, dtype=torch.float64, device="cuda") * 2 - 1
        C0 = torch.rand(m * n, dtype=torch.float64, device="cuda") * 2 - 1
        cols = [0, n // 2, n - 1]
        Bh = (B.reshape(n, m)[cols] if colmaj else B.reshape(m, n)[:, cols].t()).contiguous().cpu().numpy().ravel()
        Ch = (C0.reshape(n, m)[cols] if colmaj else C0.reshape(m, n)[:, cols].t()).contiguous().cpu().numpy().ravel()
        so, Cr = oracle.dcsrmm("col", alpha, 0, v, ci, rp, m, Bh, len(cols), m, beta, Ch, m)
        assert so == 0
        for ow in ((0, 1) if beta == 0.0 else (0,)):
            C = C0.clone()
            L.aoclsparse_mi355_set_csrmm_beta0_overwrite(ow)
            st = P.dcsrmm(P.OP_NONE, alpha, A, d, order, B, n, ld, beta, C, ld)
            L.aoclsparse_mi355_set_csrmm_beta0_overwrite(0)
            torch.cuda.synchronize()
            got = (C.reshape(n, m)[cols] if colmaj else C.reshape(m, n)[:, cols].t()).contiguous().cpu().numpy().ravel()
            ok = st == 0 and np.array_equal(got.view(np.int64), Cr.view(np.int64))
            if not ok:
                bad += 1
                print(json.dumps({"FAIL": True, "case": case, "gx": gx, "gy": gy, "nine": nine, "dofs": dofs, "thin": thin, "n": n, "colmaj": colmaj,
                                  "alpha": alpha, "beta": beta, "overwrite": ow, "status": st}), flush=True)
    del A
print(json.dumps({"cases": cases, "failures": bad}))
