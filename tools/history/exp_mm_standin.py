#!/usr/bin/env python3
"""One stand-in, csrmm with n columns, a few calls (for rocprofv3 passes).  usage: exp_mm_standin.py shell-like|flan-like [n] [row|col]"""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # (tools/history/ -> repository root)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as entry, standins
pkg = entry.load_package(); L = pkg.lib()
L.aoclsparse_mi355_set_pointer_mode(pkg.PTR_DEVICE)
dev = torch.device("cuda", 0)
name = sys.argv[1] if len(sys.argv) > 1 else "shell-like"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
colmaj = len(sys.argv) > 3 and sys.argv[3] == "col"
label, m, rp, ci, v = standins.load(name)
A = pkg.Matrix(0, m, m, rp, ci, v); d = pkg.Descr()
assert L.aoclsparse_set_mm_hint(A.h, pkg.OP_NONE, d.h, 100) == 0 and L.aoclsparse_optimize(A.h) == 0
B = torch.from_numpy(np.random.default_rng(1).uniform(-1, 1, (m, n))).to(dev)
C = torch.zeros((m, n), dtype=torch.float64, device=dev)
KID = int(os.environ["KID"]) if "KID" in os.environ else None   # KID=<0..3>: aoclsparse_dcsrmm_kid
call = (lambda: L.aoclsparse_dcsrmm(pkg.OP_NONE, 1.0, A.h, d.h, pkg.ORDER_COLUMN if colmaj else pkg.ORDER_ROW, pkg._ptr(B), n,
                                   m if colmaj else n, 0.0, pkg._ptr(C), m if colmaj else n)) if KID is None else (
    lambda: L.aoclsparse_dcsrmm_kid(pkg.OP_NONE, 1.0, A.h, d.h, pkg.ORDER_COLUMN if colmaj else pkg.ORDER_ROW, pkg._ptr(B), n,
                                    m if colmaj else n, 0.0, pkg._ptr(C), m if colmaj else n, KID))
for _ in range(3):
    assert call() == 0
torch.cuda.synchronize()
pkg.timer_start()
for _ in range(10):
    call()
print(json.dumps({"A": label, "n": n, "layout": "column-major" if colmaj else "row-major", "ms": round(pkg.timer_stop() / 10, 4),
                  "checksum": float(C.double().sum().item())}))
