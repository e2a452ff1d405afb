#!/usr/bin/env python3
"""round 4: aoclsparse_dcsr2csc (host arrays in, host arrays out) on the g^2 Laplacian: wall ms per call -- the device sort from
1 M entries on -- next to the CPU port of the reference's counting sort (oracle, one thread); bit equality of the three arrays."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # (tools/history/ -> repository root)
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
import oracle
P = entry.load_package(); L = P.lib()
out = []
for g in [int(a) for a in sys.argv[1:]] or [1000, 4096]:
    m, rp, ci, v = entry.laplace5(g)
    nnz = len(v)
    d = P.Descr()
    op, oi, ov = np.zeros(m + 1, np.int32), np.zeros(nnz, np.int32), np.zeros(nnz)
    ts = []
    for _ in range(4):
        t = time.perf_counter()
        st = L.aoclsparse_dcsr2csc(m, m, nnz, d.h, 0, P._ptr(rp), P._ptr(ci), P._ptr(v), P._ptr(oi), P._ptr(op), P._ptr(ov))
        ts.append((time.perf_counter() - t) * 1e3)
        assert st == 0
    t = time.perf_counter()
    st, cp, ri, cv = oracle.dcsr2csc(m, m, nnz, 0, 0, rp, ci, v)
    t_cpu = (time.perf_counter() - t) * 1e3
    out.append({"grid": g, "nnz": nnz, "ms": [round(x, 2) for x in ts], "cpu_port_ms": round(t_cpu, 2),
                "bit_exact": bool(np.array_equal(op, cp) and np.array_equal(oi, ri) and np.array_equal(ov, cv))})
print(json.dumps(out))
