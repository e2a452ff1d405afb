#!/usr/bin/env python3
"""round 4 (back-to-back = 100 calls between two events; laps = an event after every call, as bench.py times its legs): the 32-column row-major slab of the 1000^2 Laplacian (csrmm_tile_kernel) against the number of rows per row block
(experiment knob AOCLSPARSE_MI355_EXP_MAXROWS): ms per product, C read and overwritten, next to the 256-column product."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # (tools/history/ -> repository root)
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
pkg = entry.load_package(); L = pkg.lib(); P = pkg
m, rp, ci, v = entry.laplace5(1000)
L.aoclsparse_mi355_set_pointer_mode(P.PTR_DEVICE)
d = P.Descr()
A = P.Matrix(0, m, m, rp, ci, v)
assert L.aoclsparse_set_mm_hint(A.h, P.OP_NONE, d.h, 100) == 0 and L.aoclsparse_optimize(A.h) == 0
out = {"row_blocks": A.spmv_info().row_blocks}
def t(n, overwrite):
    L.aoclsparse_mi355_set_csrmm_beta0_overwrite(1 if overwrite else 0)
    B = torch.rand(m * n, dtype=torch.float64, device="cuda"); C = torch.zeros(m * n, dtype=torch.float64, device="cuda")
    fn = lambda: P.dcsrmm(P.OP_NONE, 1.0, A, d, P.ORDER_ROW, B, n, n, 0.0, C, n)
    best = 1e9
    for rep in range(3):
        for _ in range(10): fn()
        torch.cuda.synchronize(); P.timer_start()
        for _ in range(100): fn()
        best = min(best, P.timer_stop() / 100)
    # the bench's way: one event between consecutive calls, mean and median of 20 laps
    for _ in range(3): fn()
    L.aoclsparse_mi355_synchronize(); P.timer_mark()
    for _ in range(20):
        fn(); P.timer_mark()
    laps = np.array(P.timer_laps())
    out.setdefault("laps", {})["n%d_%s" % (n, "overwrite" if overwrite else "c_read")] = {"mean": round(float(laps.mean()), 5), "median": round(float(np.median(laps)), 5), "min": round(float(laps.min()), 5)}
    return round(best, 5)
for n in (32, 256):
    out["n%d_c_read_ms" % n] = t(n, False); out["n%d_overwrite_ms" % n] = t(n, True)
L.aoclsparse_mi355_set_csrmm_beta0_overwrite(0)
out["eff8_c_read"] = round(out["n256_c_read_ms"] / (8 * out["n32_c_read_ms"]), 4)
out["eff8_overwrite"] = round(out["n256_overwrite_ms"] / (8 * out["n32_overwrite_ms"]), 4)
print(json.dumps(out))
