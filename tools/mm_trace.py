#!/usr/bin/env python3
"""Where does a workgroup of the csrmm slab kernel (csrmm_tile_kernel, 32 columns of the 1000^2 Laplacian) spend its life?
Diagnostic: AOCLSPARSE_MI355_MM_TRACE makes the kernel dump, per row block, the 100 MHz clock at its start / after the block
table is read / when its tile loads have landed / after the barrier / at the end.   mm_trace.py [n=32]   (OVERWRITE=1: beta = 0
overwrite mode)"""
import json, os, sys
TRACE = "/tmp/mm_trace.bin"
os.environ["AOCLSPARSE_MI355_MM_TRACE"] = TRACE
if os.environ.get("OVERWRITE"):
    os.environ["AOCLSPARSE_MI355_CSRMM_BETA0_OVERWRITE"] = "1"
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as entry
pkg = entry.load_package(); L = pkg.lib()
L.aoclsparse_mi355_set_pointer_mode(pkg.PTR_DEVICE)
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
m, rp, ci, v = entry.laplace5(1000)
A = pkg.Matrix(0, m, m, rp, ci, v); d = pkg.Descr()
B = torch.from_numpy(np.random.default_rng(1).uniform(-1, 1, (m, n))).to(dev)
C = torch.zeros((m, n), dtype=torch.float64, device=dev)
for _ in range(4):
    assert L.aoclsparse_dcsrmm(pkg.OP_NONE, 1.0, A.h, d.h, pkg.ORDER_ROW, pkg._ptr(B), n, n, 0.0, pkg._ptr(C), n) == 0
torch.cuda.synchronize()
t = np.fromfile(TRACE, dtype=np.uint64).reshape(-1, 8)
t = t[t[:, 4] > 0]
st, tb, tl, ts, te = (t[:, k].astype(np.int64) for k in range(5))
t0 = st.min()
us = lambda a: a / 100.0
q = lambda a: [round(float(np.percentile(a, p)), 2) for p in (5, 25, 50, 75, 95, 100)]
rows = (t[:, 5] >> np.uint64(32)).astype(np.int64)
# concurrency: how many workgroups are alive at a time (sampled)
grid = np.linspace(t0, te.max(), 400)
alive = [(int(((st <= g) & (te > g)).sum())) for g in grid]
print(json.dumps({"what": "csrmm_tile_kernel trace, 1000^2 Laplacian, %d columns, %s" % (n, "C overwritten" if os.environ.get("OVERWRITE") else "C read"),
                  "row_blocks": int(len(t)), "kernel_span_us": float(us(te.max() - t0)),
                  "start_us_q (dispatch ramp)": q(us(st - t0)), "block_table_us_q": q(us(tb - st)), "tile_loads_us_q": q(us(tl - tb)),
                  "barrier_us_q": q(us(ts - tl)), "rows_us_q (after the barrier -> end)": q(us(te - ts)), "whole_block_us_q": q(us(te - st)),
                  "rows_per_block_q": q(rows), "us_per_row_iteration_q (rows / 32 per iteration)": q(us(te - ts) / np.maximum(1, np.ceil(rows / 32.0))),
                  "workgroups_alive_q": q(alive), "quartiles": "[5, 25, 50, 75, 95, 100] %"}))
