#!/usr/bin/env python3
"""GPU-side A/B of the SpMV planner knobs (tile, XCD order) on the bench workload; prints one line per
variant.  Interleaved rounds in ONE process (cdna_hip_programming.md, rule 24)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package()
L = pkg.lib()
g = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
m, rp, ci, v = entry.laplace5(g)
nnz = len(v)
abytes = (m + 1 + nnz) * 4 + (2 * m + nnz) * 8
x = torch.from_numpy(np.sin(0.01 * np.arange(m))).cuda()
y = torch.zeros(m, dtype=torch.float64, device="cuda")
L.aoclsparse_mi355_set_pointer_mode(pkg.PTR_DEVICE)
d = pkg.Descr()
variants = []
for tile in (512, 1024, 2048):
    for xcd in (0, 1):
        os.environ["AOCLSPARSE_MI355_SPMV_TILE"] = str(tile)
        os.environ["AOCLSPARSE_MI355_XCD_ORDER"] = str(xcd)
        A = pkg.Matrix(0, m, m, rp, ci, v)
        assert L.aoclsparse_set_mv_hint(A.h, pkg.OP_NONE, d.h, 100) == 0 and L.aoclsparse_optimize(A.h) == 0
        variants.append(("tile=%d xcd=%d" % (tile, xcd), A))
res = {n: [] for n, _ in variants}
for r in range(rounds):
    for name, A in variants:
        for _ in range(3):
            pkg.dmv(pkg.OP_NONE, 1.0, A, d, x, 0.0, y)
        pkg.timer_start()
        for _ in range(20):
            pkg.dmv(pkg.OP_NONE, 1.0, A, d, x, 0.0, y)
        res[name].append(pkg.timer_stop() / 20)
for name, _ in variants:
    t = np.array(res[name])
    print("%-18s median %.4f ms  min %.4f ms  -> %.0f GB/s (%.1f%% of 8 TB/s)" % (
        name, np.median(t), t.min(), abytes / np.median(t) / 1e6, abytes / np.median(t) / 1e6 / 80))
