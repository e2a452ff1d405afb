#!/usr/bin/env python3
"""SpMV on the mesh stand-ins (shell-like 5 dofs per node, flan-like 3) through aoclsparse_dmv after optimize: SELL-64 with
shared column lists vs plain SELL-64 (AOCLSPARSE_MI355_SELL_SHARED=0).  Diagnostic; one JSON line per matrix."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # (tools/history/ -> repository root)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as entry, oracle, standins
from bench import spmv_bytes
pkg = entry.load_package(); L = pkg.lib()
L.aoclsparse_mi355_set_pointer_mode(pkg.PTR_DEVICE)
dev = torch.device("cuda", 0)
for name in sys.argv[1:] or ["shell-like", "flan-like"]:
    label, m, rp, ci, v = standins.load(name)
    A = pkg.Matrix(0, m, m, rp, ci, v); d = pkg.Descr()
    assert L.aoclsparse_set_mv_hint(A.h, pkg.OP_NONE, d.h, 1000) == 0 and L.aoclsparse_optimize(A.h) == 0
    xh = np.random.default_rng(1).uniform(-1, 1, m)
    x = torch.from_numpy(xh).to(dev); y = torch.zeros(m, dtype=torch.float64, device=dev)
    for _ in range(5):
        assert pkg.dmv(pkg.OP_NONE, 1.0, A, d, x, 0.0, y) == 0
    torch.cuda.synchronize(); pkg.timer_start()
    for _ in range(200):
        pkg.dmv(pkg.OP_NONE, 1.0, A, d, x, 0.0, y)
    ms = pkg.timer_stop() / 200
    st, yr = oracle.dcsrmv(-1, 0, 1.0, m, len(v), v, ci, rp, xh, 0.0, np.zeros(m), nthreads=oracle.max_threads())
    b = spmv_bytes(m, m, len(v))
    print(json.dumps({"matrix": label, "kernel": pkg.Matrix.spmv_info(A).kernel, "shared_env": os.environ.get("AOCLSPARSE_MI355_SELL_SHARED"),
                      "ms": round(ms, 5), "gflops": round(2.0 * len(v) / ms / 1e6, 1), "algorithmic_GBs": round(b / ms / 1e6, 1),
                      "frac_of_8TBs": round(b / ms / 1e6 / 8000, 4), "bit_exact": bool(np.array_equal(y.cpu().numpy(), yr))}), flush=True)
