#!/usr/bin/env python3
"""Round 3: is the irregular SpMV bound by x traffic?  The two graph stand-ins and their LOCAL variants
(tools/standins.py: 70-80 % of the columns near the row), plain block order and XCD-contiguous block order
(AOCLSPARSE_MI355_XCD_ORDER=1: every XCD walks one contiguous eighth of the rows, so with locality its L2 sees one window of x).
  exp_irregular_r3.py [name ...]      one JSON line per matrix: us per call (200 back to back), kernel, bit-exactness
Run under `rocprofv3 --pmc FETCH_SIZE` (tools/exp_irregular_r3.sh) for the fabric-side bytes per launch."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # (tools/history/ -> repository root)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as entry, oracle, standins
pkg = entry.load_package(); L = pkg.lib()
L.aoclsparse_mi355_set_pointer_mode(pkg.PTR_DEVICE)
dev = torch.device("cuda", 0)
names = sys.argv[1:] or ["circuit-like", "circuit-like, local", "web-like", "web-like, local"]
for name in names:
    m, rp, ci, v = standins.ALL[name]()
    A = pkg.Matrix(0, m, m, rp, ci, v); d = pkg.Descr()
    assert L.aoclsparse_set_mv_hint(A.h, pkg.OP_NONE, d.h, 1000) == 0 and L.aoclsparse_optimize(A.h) == 0
    x = np.random.default_rng(1).uniform(-1, 1, m)
    xd, yd = torch.from_numpy(x).to(dev), torch.zeros(m, dtype=torch.float64, device=dev)
    for _ in range(20):
        pkg.dmv(pkg.OP_NONE, 1.0, A, d, xd, 0.0, yd)
    torch.cuda.synchronize()
    pkg.timer_start()
    for _ in range(200):
        pkg.dmv(pkg.OP_NONE, 1.0, A, d, xd, 0.0, yd)
    us = pkg.timer_stop() / 200 * 1e3
    so, yr = oracle.dcsrmv(-1, 0, 1.0, m, len(v), v, ci, rp, x, 0.0, np.zeros(m), nthreads=oracle.max_threads())
    inf = A.spmv_info()
    lens = np.diff(rp)
    within = lens <= max(inf.tile, 1)
    got = yd.cpu().numpy()
    print(json.dumps({"matrix": name, "m": m, "nnz": len(v), "xcd_order": os.environ.get("AOCLSPARSE_MI355_XCD_ORDER", "0"),
                      "kernel": inf.kernel, "row_blocks": inf.row_blocks, "us": round(us, 2),
                      "algorithmic_MB": round(((m + 1 + len(v)) * 4 + (2 * m + len(v)) * 8) / 1e6, 2),
                      "bit_exact_rows_within_tile": bool(np.array_equal(got[within], yr[within]))}), flush=True)
