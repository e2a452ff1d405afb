#!/usr/bin/env python3
"""round 4: block-plan statistics and timing of aoclsparse_dtrsv on the ILU(0) factors of the two shell-like stand-ins
(AOCLSPARSE_MI355_TRSV_TRACE=<file> makes the library print the plan's block / slice / level counts)."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # (tools/history/ -> repository root)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as entry, standins, oracle
pkg = entry.load_package(); L = pkg.lib()
L.aoclsparse_mi355_set_pointer_mode(pkg.PTR_DEVICE)
for name in sys.argv[1:] or ["shell-like", "shell-like, unstructured"]:
    label, m, rp, ci, v = standins.load(name)
    st, lu, dg = oracle.dilu0(m, 0, rp, ci, v)
    A = pkg.Matrix(0, m, m, rp, ci, lu)
    dl = pkg.Descr(mtype=pkg.TYPE_TRIANGULAR, fill=pkg.FILL_LOWER, diag=pkg.DIAG_UNIT)
    assert L.aoclsparse_set_sv_hint(A.h, pkg.OP_NONE, dl.h, 100) == 0 and L.aoclsparse_optimize(A.h) == 0
    b = torch.ones(m, dtype=torch.float64, device="cuda"); x = torch.zeros_like(b)
    for _ in range(3):
        assert pkg.dtrsv(pkg.OP_NONE, 1.0, A, dl, b, x) == 0
    torch.cuda.synchronize(); pkg.timer_start()
    for _ in range(10):
        pkg.dtrsv(pkg.OP_NONE, 1.0, A, dl, b, x)
    ms = pkg.timer_stop() / 10
    o = oracle.dcsr_optimize(m, m, len(lu), 0, rp, ci, lu)
    lens = o["idiag"] - rp[:-1]
    print(json.dumps({"matrix": label, "m": m, "row_levels": A.trsv_levels(pkg.FILL_LOWER), "ms": round(ms, 4),
                      "strict_lower_row_len": {"mean": float(lens.mean()), "max": int(lens.max()),
                                               "hist_0_8_16_24_32_48": np.histogram(lens, [0, 8, 16, 24, 32, 48, 1 << 30])[0].tolist()}}), flush=True)
