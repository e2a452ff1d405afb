#!/usr/bin/env python3
"""TRSV schedule timings on the config-5 factors (diagnostic; one JSON line per measurement)."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # (tools/history/ -> repository root)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as entry, oracle, standins
from bench import timed_laps
pkg = entry.load_package(); L = pkg.lib()
L.aoclsparse_mi355_set_pointer_mode(pkg.PTR_DEVICE)
dev = torch.device("cuda", 0)
cases = [("lap1000", lambda: entry.laplace5(1000))]
if "--small" not in sys.argv:
    cases.append(("shell", standins.shell_like))
for title, gen in cases:
    m, rp, ci, v = gen()
    st, lu, dg = oracle.dilu0(m, 0, rp, ci, v)
    A = pkg.Matrix(0, m, m, rp, ci, lu)
    o = oracle.dcsr_optimize(m, m, len(lu), 0, rp, ci, lu)
    lv = A.trsv_levels(pkg.FILL_LOWER)
    b = np.random.default_rng(2).uniform(-1, 1, m)
    bd, xd = torch.from_numpy(b).to(dev), torch.zeros(m, dtype=torch.float64, device=dev)
    variants = [("l", pkg.FILL_LOWER, pkg.OP_NONE, True, o["idiag"])]
    if "--all" in sys.argv:
        variants += [("u", pkg.FILL_UPPER, pkg.OP_NONE, False, o["iurow"]), ("lt", pkg.FILL_LOWER, pkg.OP_TRANSPOSE, True, o["idiag"]),
                     ("ut", pkg.FILL_UPPER, pkg.OP_TRANSPOSE, False, o["iurow"])]
    for kind, fill, op, unit, iend in variants:
        dl = pkg.Descr(mtype=pkg.TYPE_TRIANGULAR, fill=fill, diag=pkg.DIAG_UNIT if unit else pkg.DIAG_NON_UNIT)
        assert L.aoclsparse_set_sv_hint(A.h, op, dl.h, 100) == 0 and L.aoclsparse_optimize(A.h) == 0
        st, xr = oracle.dtrsv(kind, 1.0, m, 0, lu, ci, rp, iend, b, unit)
        for kid in (2, -1):  # schedule: lane per position, automatic
            assert L.aoclsparse_mi355_set_trsv_schedule(kid) == 0
            lp = timed_laps(pkg, lambda: pkg.dtrsv(op, 1.0, A, dl, bd, xd), 10, 2)
            assert L.aoclsparse_mi355_set_trsv_schedule(-1) == 0
            torch.cuda.synchronize()
            print(json.dumps({"sys": title, "kind": kind, "levels": lv, "schedule": kid, "env_sf": os.environ.get("AOCLSPARSE_MI355_TRSV_SYNCFREE"),
                              "env_blocks": os.environ.get("AOCLSPARSE_MI355_TRSV_BLOCKS"), "ms_median": float(np.median(lp)),
                              "us_per_level": float(np.median(lp)) * 1e3 / lv,
                              "bit_exact": bool(np.array_equal(xd.cpu().numpy(), xr, equal_nan=True))}), flush=True)
