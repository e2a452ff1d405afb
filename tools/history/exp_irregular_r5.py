#!/usr/bin/env python3
"""aoclsparse_dmv on the two power-law stand-ins of BASELINE config 3 (and their local variants): microseconds per call (200 calls
back to back between two events) + the parity verdicts of bench.py's mix leg.  One JSON line per matrix; the environment (the
round's experiment switches, if any) is echoed so that runs of different builds / switches can sit in one file."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # (tools/history/ -> repository root)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as entry, standins, oracle
pkg = entry.load_package(); L = pkg.lib()
L.aoclsparse_mi355_set_pointer_mode(pkg.PTR_DEVICE)
dev = torch.device("cuda", 0)
args = sys.argv[1:]
if args and args[0].startswith("--kernel="):  # --kernel=merge | adaptive: force one CSR kernel (default: the automatic choice)
    assert L.aoclsparse_mi355_set_option(pkg.OPTION_SPMV_KERNEL, {"adaptive": 1, "merge": 2}[args[0].split("=")[1]]) == 0
    args = args[1:]
names = args or ["circuit-like", "web-like"]
env = {k: v for k, v in os.environ.items() if k.startswith("AOCLSPARSE_MI355_")}
for name in names:
    label, m, rp, ci, v = standins.load(name)
    nz = len(v)
    A = pkg.Matrix(0, m, m, rp, ci, v); d = pkg.Descr()
    assert L.aoclsparse_set_mv_hint(A.h, pkg.OP_NONE, d.h, 1000) == 0 and L.aoclsparse_optimize(A.h) == 0
    inf = A.spmv_info()
    xr = np.random.default_rng(1).uniform(-1, 1, m)
    x = torch.from_numpy(xr).to(dev); y = torch.zeros(m, dtype=torch.float64, device=dev)
    best = []
    for rep in range(3):
        for _ in range(20):
            pkg.dmv(pkg.OP_NONE, 1.0, A, d, x, 0.0, y)
        torch.cuda.synchronize()
        pkg.timer_start()
        for _ in range(200):
            pkg.dmv(pkg.OP_NONE, 1.0, A, d, x, 0.0, y)
        best.append(pkg.timer_stop() / 200 * 1e3)
    so, yr = oracle.dcsrmv(-1, 0, 1.0, m, nz, v, ci, rp, xr, 0.0, np.zeros(m), nthreads=oracle.max_threads())
    got = y.cpu().numpy()
    lens = np.diff(rp)
    short = lens < inf.tree_min if inf.tree_min > 0 else np.ones(m, bool)
    scale = np.zeros(m); nzr = lens > 0
    scale[nzr] = np.add.reduceat(np.abs(v * xr[ci]), rp[:-1][nzr])
    bound = (2 * np.ceil(np.log2(np.maximum(lens, 2))) + 4 + lens / 256.0) * np.finfo(np.float64).eps * scale
    b = (m + 1 + nz) * 4 + (2 * m + nz) * 8
    print(json.dumps({"matrix": label, "env": env, "kernel": inf.kernel, "tile": inf.tile, "row_blocks": inf.row_blocks,
                      "us": [round(t, 2) for t in best], "frac_of_8TBs": round(b / (min(best) * 1e-6) / 8e12, 4),
                      "short_rows_bit_exact": bool(np.array_equal(got[short], yr[short])),
                      "all_rows_within_bound": bool(np.all(np.abs(got - yr) <= bound + 1e-300)),
                      "rows_on_the_tree": int((~short).sum())}), flush=True)
