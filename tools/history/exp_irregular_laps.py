#!/usr/bin/env python3
"""Irregular SpMV (BASELINE config 3 stand-ins) through aoclsparse_dmv after optimize: median time per call (event laps,
~3 us of launch/event overhead included) and bit-exactness against the oracle.  Diagnostic; one JSON line per matrix."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # (tools/history/ -> repository root)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as entry, oracle, standins
from bench import timed_laps
pkg = entry.load_package(); L = pkg.lib()
L.aoclsparse_mi355_set_pointer_mode(pkg.PTR_DEVICE)
dev = torch.device("cuda", 0)
for name, gen in (("circuit-like", standins.circuit_like), ("web-like", standins.web_like)):
    m, rp, ci, v = gen()
    A = pkg.Matrix(0, m, m, rp, ci, v); d = pkg.Descr()
    assert L.aoclsparse_set_mv_hint(A.h, pkg.OP_NONE, d.h, 1000) == 0 and L.aoclsparse_optimize(A.h) == 0
    xh = np.random.default_rng(1).uniform(-1, 1, m)
    x = torch.from_numpy(xh).to(dev); y = torch.zeros(m, dtype=torch.float64, device=dev)
    lp = timed_laps(pkg, lambda: pkg.dmv(pkg.OP_NONE, 1.0, A, d, x, 0.0, y), 300, 20)
    torch.cuda.synchronize()
    st, yr = oracle.dcsrmv(0, 0, 1.0, m, len(v), v, ci, rp, xh, 0.0, np.zeros(m))
    got = y.cpu().numpy()
    print(json.dumps({"matrix": name, "env": {k: os.environ[k] for k in os.environ if k.startswith("AOCLSPARSE_MI355_SPMV")},
                      "us_median": round(float(np.median(lp)) * 1e3, 2), "us_q1": round(float(np.percentile(lp, 25)) * 1e3, 2),
                      "us_min": round(float(np.min(lp)) * 1e3, 2), "rows_differing": int((got != yr).sum())}), flush=True)
