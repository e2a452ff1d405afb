#!/usr/bin/env python3
"""round 4: aoclsparse_sp2m (C = A * A, full computation, host arrays in / host arrays out) against the CPU restatement of the
reference's two-stage Gustavson (oracle/, one thread) -- wall time per call, and bit equality of row_ptr / col_ind / val."""
import ctypes, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # (tools/history/ -> repository root)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as entry
import oracle, standins
pkg = entry.load_package(); L = pkg.lib()
which = sys.argv[1] if len(sys.argv) > 1 else "laplace"
size = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
opa = pkg.OP_TRANSPOSE if len(sys.argv) > 3 and sys.argv[3] == "T" else pkg.OP_NONE  # A^T * A (cached transpose) when "T"
if which == "laplace":
    m, rp, ci, v = entry.laplace5(size)
elif which == "shell":  # 5-dof shell mesh of `size` rows (35 entries per row): products per row ~ 1,225
    m, rp, ci, v = standins.shell_like(n=size)
elif which == "circuit":
    m, rp, ci, v = standins.circuit_like()
else:
    from util import random_csr
    m = size
    rp, ci, v = random_csr(3, m, m, lambda r, i: r.integers(0, 30))
A = pkg.Matrix(0, m, m, rp, ci, v)
d = pkg.Descr()
def run():
    C = ctypes.c_void_p()
    t = time.perf_counter()
    st = L.aoclsparse_sp2m(opa, d.h, A.h, pkg.OP_NONE, d.h, A.h, pkg.STAGE_FULL, ctypes.byref(C))
    dt = time.perf_counter() - t
    assert st == 0
    return C, dt
C, _ = run(); L.aoclsparse_destroy(ctypes.byref(C))
ts = []
for _ in range(3):
    C, dt = run(); ts.append(dt)
    if _ < 2: L.aoclsparse_destroy(ctypes.byref(C))
h = pkg.Matrix.from_handle(C); e = h.export()
t = time.perf_counter()
if opa == pkg.OP_NONE:
    so, pc, ic, vc = oracle.dcsr2m(m, m, 0, rp, ci, v, 0, rp, ci, v)
else:
    st_, cp_, ri_, cv_ = oracle.dcsr2csc(m, m, len(v), 0, 0, rp, ci, v)
    so, pc, ic, vc = oracle.dcsr2m(m, m, 0, cp_, ri_.astype(np.int32), cv_, 0, rp, ci, v)
t_cpu = time.perf_counter() - t
print(json.dumps({"matrix": which, "opA": "T" if opa != pkg.OP_NONE else "N", "m": m, "nnz_a": int(len(v)), "nnz_c": int(e["nnz"]), "sp2m_wall_ms": [round(x * 1e3, 2) for x in ts],
                  "cpu_port_1_thread_ms": round(t_cpu * 1e3, 2),
                  "bit_exact": bool(np.array_equal(e["row_ptr"], pc) and np.array_equal(e["col_ind"], ic) and np.array_equal(e["val"], vc))}))
