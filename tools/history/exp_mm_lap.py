#!/usr/bin/env python3
"""csrmm on the 1000^2 5-point Laplacian, n columns, straight through aoclsparse_dcsrmm (device pointers):
  exp_mm_lap.py [n] [row|col]      (KID=<0..3>: through aoclsparse_dcsrmm_kid)"""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # (tools/history/ -> repository root)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as entry
pkg = entry.load_package(); L = pkg.lib()
L.aoclsparse_mi355_set_pointer_mode(pkg.PTR_DEVICE)
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
colmaj = len(sys.argv) > 2 and sys.argv[2] == "col"
m, rp, ci, v = entry.laplace5(1000)
A = pkg.Matrix(0, m, m, rp, ci, v); d = pkg.Descr()
B = torch.from_numpy(np.random.default_rng(1).uniform(-1, 1, (m, n))).to(dev)
C = torch.zeros((m, n), dtype=torch.float64, device=dev)
KID = int(os.environ["KID"]) if "KID" in os.environ else None
if KID is None:
    call = lambda: L.aoclsparse_dcsrmm(pkg.OP_NONE, 1.0, A.h, d.h, pkg.ORDER_COLUMN if colmaj else pkg.ORDER_ROW, pkg._ptr(B), n,
                                      m if colmaj else n, 0.0, pkg._ptr(C), m if colmaj else n)
else:
    call = lambda: L.aoclsparse_dcsrmm_kid(pkg.OP_NONE, 1.0, A.h, d.h, pkg.ORDER_COLUMN if colmaj else pkg.ORDER_ROW, pkg._ptr(B), n,
                                          m if colmaj else n, 0.0, pkg._ptr(C), m if colmaj else n, KID)
for _ in range(3):
    assert call() == 0
torch.cuda.synchronize()
pkg.timer_start()
for _ in range(20):
    call()
print(json.dumps({"A": "laplace5 1000^2", "n": n, "layout": "column-major" if colmaj else "row-major", "kid": KID, "ms": round(pkg.timer_stop() / 20, 4)}))
