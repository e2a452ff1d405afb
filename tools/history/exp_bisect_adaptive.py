#!/usr/bin/env python3
"""round 4: raw aoclsparse_dcsrmv (device arrays, CSR-Adaptive kernel) on the 4096^2 Laplacian through a given build of the
library, loaded directly with ctypes (only the reference's own ABI is used, so old builds work): ms per product."""
import ctypes, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # (tools/history/ -> repository root)
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
so = sys.argv[1]
L = ctypes.CDLL(so)
m, rp, ci, v = entry.laplace5(4096)
nnz = len(v)
descr = ctypes.c_void_p()
assert L.aoclsparse_create_mat_descr(ctypes.byref(descr)) == 0
x = torch.from_numpy(np.sin(0.01 * np.arange(m))).cuda(); y = torch.zeros(m, dtype=torch.float64, device="cuda")
drp, dci, dv = (torch.from_numpy(a).cuda() for a in (rp, ci, v))
a, b = ctypes.c_double(1.0), ctypes.c_double(0.0)
P = ctypes.c_void_p
L.aoclsparse_dcsrmv.argtypes = [ctypes.c_int, P, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, P, P, P, P, P, P, P]
def call():
    st = L.aoclsparse_dcsrmv(111, ctypes.byref(a), m, m, nnz, dv.data_ptr(), dci.data_ptr(), drp.data_ptr(), descr, x.data_ptr(), ctypes.byref(b), y.data_ptr())
    assert st == 0, st
for _ in range(10): call()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e9
for rep in range(5):
    e0.record()
    for _ in range(50): call()
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 50)
print(json.dumps({"lib": os.path.basename(so), "raw_dcsrmv_ms": round(best, 5), "checksum": float(y.sum().item())}))
