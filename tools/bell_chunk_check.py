#!/usr/bin/env python3
"""Blocked-ELL MFMA csrmm: the XCD chunk build_bell's model picks against forced chunks (AOCLSPARSE_MI355_BELL_XCD_CHUNK, read at analysis
time), block-dense stand-ins.  python tools/bell_chunk_check.py [edges=32,40] [chunks=auto,1,2,4,5,8,16,32,-1] [cols=256] [row|col] [keep=1.0]
One JSON line per (edge, chunk); auto = the model's choice, -1 = the lattice sweep, 0 = launch order without a list.  Every product is checked against chunk 1's bits."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as entry, standins
pkg = entry.load_package(); L = pkg.lib()
edges = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "32,40").split(",")]
chunks = [x for x in (sys.argv[2] if len(sys.argv) > 2 else "auto,1,2,4,5,8,16,32,-1").split(",")]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 256
order = pkg.ORDER_COLUMN if len(sys.argv) > 4 and sys.argv[4] == "col" else pkg.ORDER_ROW
keep = float(sys.argv[5]) if len(sys.argv) > 5 else 1.0
L.aoclsparse_mi355_set_pointer_mode(pkg.PTR_DEVICE)
for e in edges:
    m, rp, ci, v = standins.block_dense(e, e, e, keep=keep)
    ld = m if order == pkg.ORDER_COLUMN else n
    B = torch.rand(m * n, dtype=torch.float64, device="cuda") * 2 - 1
    C = torch.zeros(m * n, dtype=torch.float64, device="cuda")
    ref = None
    for ch in ["1"] + [c for c in chunks if c != "1"]:
        if ch != "auto":
            os.environ["AOCLSPARSE_MI355_BELL_XCD_CHUNK"] = ch
        else:
            os.environ.pop("AOCLSPARSE_MI355_BELL_XCD_CHUNK", None)
        A = pkg.Matrix(0, m, m, rp, ci, v); d = pkg.Descr()
        assert L.aoclsparse_set_mm_hint(A.h, pkg.OP_NONE, d.h, 100) == 0 and L.aoclsparse_optimize(A.h) == 0
        info = A.spmv_info()
        rec = {"edge": e, "m": m, "cols": n, "order": "col" if order == pkg.ORDER_COLUMN else "row", "keep": keep, "forced": ch,
               "xcd_chunk": info.mm_bell_xcd_chunk, "model_fetches": info.mm_bell_model_fetches_permille / 1000.0,
               "model_fetches_launch_order": info.mm_bell_model_fetches_launch_order_permille / 1000.0,
               "lattice": [info.mm_bell_lattice_line, info.mm_bell_lattice_lines], "region": [info.mm_bell_region_a, info.mm_bell_region_b]}
        for ow in (0, 1):
            L.aoclsparse_mi355_set_csrmm_beta0_overwrite(ow)
            for _ in range(3):
                assert pkg.dcsrmm(pkg.OP_NONE, 1.0, A, d, order, B, n, ld, 0.0, C, ld) == 0
            torch.cuda.synchronize(); pkg.timer_start()
            for _ in range(10):
                pkg.dcsrmm(pkg.OP_NONE, 1.0, A, d, order, B, n, ld, 0.0, C, ld)
            ms = pkg.timer_stop() / 10
            rec["overwrite_ms" if ow else "c_read_ms"] = round(ms, 4)
            rec["overwrite_tflops" if ow else "c_read_tflops"] = round(2.0 * len(v) * n / ms / 1e9, 2)
        L.aoclsparse_mi355_set_csrmm_beta0_overwrite(0)
        if ref is None:
            ref = C.clone()
            rec["same_bits_as_chunk_1"] = True
        else:
            rec["same_bits_as_chunk_1"] = bool(torch.equal(C.view(torch.int64), ref.view(torch.int64)))
        print(json.dumps(rec), flush=True)
        del A
    del B, C, ref
