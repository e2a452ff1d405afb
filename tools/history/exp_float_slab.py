import json, os, sys
import numpy as np, torch
sys.path.insert(0, "/root/repo")
import __graft_entry__ as entry
pkg = entry.load_package(); L = pkg.lib()
m, rp, ci, v = entry.laplace5(1000)
L.aoclsparse_mi355_set_pointer_mode(pkg.PTR_DEVICE)
d = pkg.Descr()
A = pkg.Matrix(0, m, m, rp, ci, v.astype(np.float32))
assert L.aoclsparse_set_mm_hint(A.h, pkg.OP_NONE, d.h, 100) == 0 and L.aoclsparse_optimize(A.h) == 0
out = {"tile": A.spmv_info().tile}
for n in (32, 64):
    B = torch.rand(m * n, dtype=torch.float32, device="cuda"); C = torch.zeros(m * n, dtype=torch.float32, device="cuda")
    fn = lambda: pkg.scsrmm(pkg.OP_NONE, 1.0, A, d, pkg.ORDER_ROW, B, n, n, 0.0, C, n)
    for _ in range(10): fn()
    torch.cuda.synchronize(); pkg.timer_start()
    for _ in range(100): fn()
    out["row_n%d_ms" % n] = round(pkg.timer_stop() / 100, 5)
print(json.dumps(out))
