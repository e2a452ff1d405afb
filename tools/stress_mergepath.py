#!/usr/bin/env python3
"""round 5: the one-launch merge-path kernel under the conditions a decoupled look-back has to survive: far more tiles than the chip
holds at once (here ~15,000 of 1,024 items against 2,048 resident workgroups), rows that cross hundreds of tiles, a competing
stream that keeps every CU busy (uneven load), repeated launches (epoch tags), and every word of y checked every time.
  python3 tools/stress_mergepath.py [rows=3000000] [long_rows=6] [length=900000] [repeats=40]"""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle
from util import pkg
P = pkg(); L = P.lib()
m = int(sys.argv[1]) if len(sys.argv) > 1 else 3000000
nlong = int(sys.argv[2]) if len(sys.argv) > 2 else 6
length = int(sys.argv[3]) if len(sys.argv) > 3 else 900000
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 40
rng = np.random.default_rng(11)
longs = np.sort(rng.choice(m, size=nlong, replace=False))
lens = np.full(m, 3, np.int64); lens[0] = lens[-1] = 2
lens[longs] = length
rp = np.zeros(m + 1, np.int64); np.cumsum(lens, out=rp[1:])
nnz = int(rp[m])
ci = np.empty(nnz, np.int32)
i = np.arange(m, dtype=np.int64)
# tridiagonal part, vectorised
tri_rows = np.setdiff1d(i, longs)
for k, off in enumerate((-1, 0, 1)):
    pass
pos = rp[:-1]
mid = (lens == 3)
ci[pos[mid]] = (i[mid] - 1).astype(np.int32); ci[pos[mid] + 1] = i[mid].astype(np.int32); ci[pos[mid] + 2] = (i[mid] + 1).astype(np.int32)
for r in (0, m - 1):
    if lens[r] == 2:
        ci[pos[r]:pos[r] + 2] = [0, 1] if r == 0 else [m - 2, m - 1]
for r in longs:
    c = np.sort(rng.choice(m, size=length, replace=False)).astype(np.int32)
    ci[pos[r]:pos[r] + length] = c
v = rng.uniform(-1, 1, nnz)
rp32 = rp.astype(np.int32)
x = rng.uniform(-1, 1, m)
L.aoclsparse_mi355_set_pointer_mode(P.PTR_DEVICE)
assert L.aoclsparse_mi355_set_option(P.OPTION_SELL, 0) == 0
A = P.Matrix(0, m, m, rp32, ci, v); d = P.Descr()
assert L.aoclsparse_set_mv_hint(A.h, P.OP_NONE, d.h, 100) == 0 and L.aoclsparse_optimize(A.h) == 0
info = A.spmv_info()
assert info.kernel == 2, info.kernel
so, yr = oracle.dcsrmv(0, 0, 1.0, m, nnz, v, ci, rp32, x, 0.0, np.zeros(m), nthreads=oracle.max_threads())
scale = np.add.reduceat(np.abs(v * x[ci]), rp[:-1])
bound = (2 * np.ceil(np.log2(np.maximum(lens, 2))) + 8 + lens / 256.0 + lens / 1024.0) * 2.0 ** -52 * scale
xd = torch.from_numpy(x).cuda()
side = torch.cuda.Stream()
noise = torch.rand(64 * 1024 * 1024, device="cuda")
first, worst, differ = None, 0.0, 0
t0 = time.time()
for rep in range(reps):
    yd = torch.full((m,), float("nan"), dtype=torch.float64, device="cuda")
    if rep % 2 == 1:  # a competing stream keeps the CUs busy while the product runs
        with torch.cuda.stream(side):
            for _ in range(6):
                noise.mul_(1.0000001)
    assert P.dmv(P.OP_NONE, 1.0, A, d, xd, 0.0, yd) == 0
    torch.cuda.synchronize()
    y = yd.cpu().numpy()
    err = np.abs(y - yr)
    assert not np.isnan(y).any(), "NaN in y: a look-back expired or a row was never written (rep %d)" % rep
    assert np.all(err <= bound + 1e-300), "rep %d: %d rows outside the bound" % (rep, int((err > bound).sum()))
    worst = max(worst, float((err / (bound + 1e-300)).max()))
    if first is None:
        first = y
    else:
        differ += int(np.sum(first != y))  # the order of additions depends on the tiling only: every launch gives the same bits
assert differ == 0, differ
print(json.dumps({"tool": "stress_mergepath", "m": m, "nnz": nnz, "tiles": int((m + nnz) // 1024 + 1), "long_rows": nlong, "length": length,
                  "tiles_per_long_row": int(length // 1024), "launches": reps, "with_competing_stream": reps // 2,
                  "every_launch_bit_identical": differ == 0, "worst_err_over_bound": round(worst, 4), "seconds": round(time.time() - t0, 1)}))
