#!/usr/bin/env python3
"""Round 3, feasibility only: what would SELL-64 with rows sorted by length inside windows of sigma rows (SELL-C-sigma) + a
separate kernel for the long rows give on the power-law stand-ins?  Built from EXISTING entry points on permuted copies of the
matrix (no new kernel): (a) the matrix as it is (CSR-Adaptive, what optimize picks today); (b) short part: rows longer than T
emptied, rows stably sorted by length inside windows of sigma, SELL-64 forced (AOCLSPARSE_MI355_SELL=1 must be set by the
caller for this process); (c) long part: the rows longer than T as a matrix of their own (CSR-Adaptive).  y of (b) is the
permuted vector, i.e. the scattered store of a real implementation is NOT in the figure.
(b) and (c) run in separate processes: the SELL switch is read once.
usage: AOCLSPARSE_MI355_SELL=1 exp_r3_sellsigma.py short <name> [T=32] [sigma=4096];  exp_r3_sellsigma.py long <name> [T]"""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # (tools/history/ -> repository root)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as entry, standins
pkg = entry.load_package(); L = pkg.lib()
L.aoclsparse_mi355_set_pointer_mode(pkg.PTR_DEVICE)
dev = torch.device("cuda", 0)
part = sys.argv[1]
name = sys.argv[2] if len(sys.argv) > 2 else "web-like"
T = int(sys.argv[3]) if len(sys.argv) > 3 else 32
sigma = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
label, m, rp, ci, v = standins.load(name)
rp = rp.astype(np.int64); lens = np.diff(rp)


def sub(rows):
    """CSR of the given rows (in that order)"""
    l = lens[rows]; nrp = np.zeros(len(rows) + 1, np.int64); nrp[1:] = np.cumsum(l)
    idx = np.concatenate([np.arange(rp[r], rp[r + 1]) for r in rows]) if len(rows) < 100000 else None
    if idx is None:
        idx = np.repeat(rp[rows] - nrp[:-1], l) + np.arange(nrp[-1])
    return nrp.astype(np.int32), ci[idx].astype(np.int32), v[idx]


def timed(mm, n, rp_, ci_, v_, hint=True):
    A = pkg.Matrix(0, mm, n, rp_, ci_, v_); d = pkg.Descr()
    if hint:
        assert L.aoclsparse_set_mv_hint(A.h, pkg.OP_NONE, d.h, 1000) == 0
    assert L.aoclsparse_optimize(A.h) == 0
    x = torch.from_numpy(np.random.default_rng(1).uniform(-1, 1, n)).to(dev); y = torch.zeros(mm, dtype=torch.float64, device=dev)
    for _ in range(5):
        assert pkg.dmv(pkg.OP_NONE, 1.0, A, d, x, 0.0, y) == 0
    torch.cuda.synchronize(); pkg.timer_start()
    for _ in range(200):
        pkg.dmv(pkg.OP_NONE, 1.0, A, d, x, 0.0, y)
    us = pkg.timer_stop() / 200 * 1e3
    return round(us, 2), int(A.spmv_info().kernel)


out = {"matrix": label, "T": T, "sigma": sigma}
long_rows = np.nonzero(lens > T)[0]
if part == "short":
  # (b) short part, sorted inside windows
  keep = lens.copy(); keep[long_rows] = 0
  order = np.concatenate([w0 + np.argsort(-keep[w0:w0 + sigma], kind="stable") for w0 in range(0, m, sigma)])
  l2 = keep[order]; nrp = np.zeros(m + 1, np.int64); nrp[1:] = np.cumsum(l2)
  idx = np.repeat(rp[order] - nrp[:-1], l2) + np.arange(nrp[-1])
  out["short_part_sell_sigma_us"], out["short_kernel"] = timed(m, m, nrp.astype(np.int32), ci[idx].astype(np.int32), v[idx])
  out["short_nnz"] = int(nrp[-1])
else:
  # (c) long rows alone
  lrp, lci, lv = sub(long_rows)
  out["long_rows"] = int(len(long_rows)); out["long_nnz"] = int(len(lv))
  out["long_part_us"], out["long_kernel"] = timed(len(long_rows), m, lrp, lci, lv)
print(json.dumps(out), flush=True)
