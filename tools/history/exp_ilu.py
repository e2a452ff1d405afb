#!/usr/bin/env python3
"""aoclsparse_dilu_smoother on the shell-like stand-in: first call (GPU ILU(0) factorisation + analysis of both factors) and
apply (L then U solve).  Diagnostic."""
import ctypes, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # (tools/history/ -> repository root)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as entry, oracle, standins
from bench import timed_laps
pkg = entry.load_package(); L = pkg.lib()
L.aoclsparse_mi355_set_pointer_mode(pkg.PTR_DEVICE)
dev = torch.device("cuda", 0)
for name, gen in (("shell-like", standins.shell_like), ("laplace5 1000^2", lambda: entry.laplace5(1000))):
    m, rp, ci, v = gen()
    A = pkg.Matrix(0, m, m, rp, ci, v.copy()); d0 = pkg.Descr()
    bh = np.random.default_rng(3).uniform(-1, 1, m)
    bd = torch.from_numpy(bh).to(dev); xi = torch.zeros(m, dtype=torch.float64, device=dev)
    pv = ctypes.c_void_p()
    torch.cuda.synchronize(); t0 = time.time()
    assert L.aoclsparse_dilu_smoother(pkg.OP_NONE, A.h, d0.h, ctypes.byref(pv), None, pkg._ptr(xi), pkg._ptr(bd)) == 0
    torch.cuda.synchronize(); t_first = time.time() - t0
    lp = timed_laps(pkg, lambda: L.aoclsparse_dilu_smoother(pkg.OP_NONE, A.h, d0.h, ctypes.byref(pv), None, pkg._ptr(xi), pkg._ptr(bd)), 10, 2)
    torch.cuda.synchronize()
    t0 = time.time(); so, lu, dg = oracle.dilu0(m, 0, rp, ci, v); t_fac = time.time() - t0
    t0 = time.time(); so, xr = oracle.dilu_solve(m, 0, dg, lu, rp, ci, bh); t_app = time.time() - t0
    print(json.dumps({"op": "aoclsparse_dilu_smoother", "system": name, "m": m, "first_call_s": round(t_first, 3),
                      "apply_ms": round(float(np.median(lp)), 3), "cpu_factorise_s": round(t_fac, 3), "cpu_apply_ms": round(t_app * 1e3, 2),
                      "x_bit_exact": bool(np.array_equal(xi.cpu().numpy(), xr)), "blocks_env": os.environ.get("AOCLSPARSE_MI355_TRSV_BLOCKS")}), flush=True)
