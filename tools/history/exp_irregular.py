#!/usr/bin/env python3
"""Experiment: where the time goes on the power-law stand-ins (run under rocprofv3 --kernel-trace --stats)."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import __graft_entry__ as entry
import standins
from bench import spmv_bytes
pkg = entry.load_package(); L = pkg.lib(); dev = torch.device("cuda", 0)
L.aoclsparse_mi355_set_pointer_mode(pkg.PTR_DEVICE)
d0 = pkg.Descr()
for name in sys.argv[1:] or ["web-like", "circuit-like"]:
    m, rp, ci, v = standins.ALL[name](); nnz = len(v)
    lens = np.diff(rp)
    x = torch.from_numpy(np.random.default_rng(1).uniform(-1, 1, m)).to(dev); y = torch.zeros(m, dtype=torch.float64, device=dev)
    A = pkg.Matrix(0, m, m, rp, ci, v)  # AOCLSPARSE_MI355_SPMV_KERNEL=merge|adaptive picks the kernel
    assert L.aoclsparse_set_mv_hint(A.h, pkg.OP_NONE, d0.h, 100) == 0 and L.aoclsparse_optimize(A.h) == 0
    info = A.spmv_info()
    for _ in range(5): pkg.dmv(pkg.OP_NONE, 1.0, A, d0, x, 0.0, y)
    torch.cuda.synchronize(); pkg.timer_start()
    for _ in range(50): pkg.dmv(pkg.OP_NONE, 1.0, A, d0, x, 0.0, y)
    ms = pkg.timer_stop() / 50
    print(json.dumps(dict(matrix=name, m=m, nnz=nnz, kernel=info.kernel, order=info.order, tile=info.tile, blocks=info.row_blocks,
                          long_rows=info.long_rows, max_row=int(lens.max()), p99=int(np.percentile(lens, 99)), ms=round(ms, 5),
                          ideal_us=round(spmv_bytes(m, m, nnz) / 6.2e6, 2))), flush=True)
