#!/usr/bin/env python3
"""trsm (several right-hand sides) on the shell-like ILU(0) factor: block kernel with one grid column per right-hand side vs the
lane-per-position kernel (AOCLSPARSE_MI355_TRSV_BLOCKS=0).  Diagnostic; one JSON line per (triangle, columns)."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # (tools/history/ -> repository root)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as entry, oracle, standins
from bench import timed_laps
pkg = entry.load_package(); L = pkg.lib()
L.aoclsparse_mi355_set_pointer_mode(pkg.PTR_DEVICE)
dev = torch.device("cuda", 0)
m, rp, ci, v = standins.shell_like()
st, lu, dg = oracle.dilu0(m, 0, rp, ci, v)
A = pkg.Matrix(0, m, m, rp, ci, lu)
o = oracle.dcsr_optimize(m, m, len(lu), 0, rp, ci, lu)
rng = np.random.default_rng(2)
for kind, fill, unit, iend in (("l", pkg.FILL_LOWER, True, o["idiag"]), ("u", pkg.FILL_UPPER, False, o["iurow"])):
    d = pkg.Descr(mtype=pkg.TYPE_TRIANGULAR, fill=fill, diag=pkg.DIAG_UNIT if unit else pkg.DIAG_NON_UNIT)
    for n in (1, 4, 8, 32):
        Bm = rng.uniform(-1, 1, (n, m))
        Bd = torch.from_numpy(Bm).to(dev); Xd = torch.zeros_like(Bd)
        call = lambda: L.aoclsparse_dtrsm(pkg.OP_NONE, 1.0, A.h, d.h, pkg.ORDER_COLUMN, pkg._ptr(Bd), n, m, pkg._ptr(Xd), m)
        assert call() == 0
        lp = timed_laps(pkg, call, 6, 2)
        torch.cuda.synchronize()
        st, xr = oracle.dtrsv(kind, 1.0, m, 0, lu, ci, rp, iend, Bm[n - 1], unit)
        print(json.dumps({"factor": "shell-like ILU(0)", "triangle": kind, "columns": n, "blocks_env": os.environ.get("AOCLSPARSE_MI355_TRSV_BLOCKS"),
                          "ms_median": round(float(np.median(lp)), 3), "ms_per_column": round(float(np.median(lp)) / n, 3),
                          "last_column_bit_exact": bool(np.array_equal(Xd[n - 1].cpu().numpy(), xr))}), flush=True)
