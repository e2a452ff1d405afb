#!/usr/bin/env python3
"""Where does the CSR-Adaptive SpMV kernel spend its time on the short irregular matrices (BASELINE config 3: circuit-like,
web-like)?  Diagnostic: AOCLSPARSE_MI355_SPMV_TRACE makes csr_adaptive_kernel dump, per row block, the 100 MHz clock at its
start / after the block table is read / when its tile is in LDS / after the barrier / at the end.  One JSON line per matrix."""
import json, os, sys
TRACE = "/tmp/spmv_trace.bin"
os.environ["AOCLSPARSE_MI355_SPMV_TRACE"] = TRACE
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as entry, standins
pkg = entry.load_package(); L = pkg.lib()
L.aoclsparse_mi355_set_pointer_mode(pkg.PTR_DEVICE)
dev = torch.device("cuda", 0)
q = lambda a: [round(float(np.percentile(a, p)), 2) for p in (5, 25, 50, 75, 95, 100)]
for name, gen in (("circuit-like", standins.circuit_like), ("web-like", standins.web_like)):
    m, rp, ci, v = gen()
    A = pkg.Matrix(0, m, m, rp, ci, v)
    d = pkg.Descr()
    assert L.aoclsparse_set_mv_hint(A.h, pkg.OP_NONE, d.h, 1000) == 0 and L.aoclsparse_optimize(A.h) == 0
    x = torch.from_numpy(np.random.default_rng(1).uniform(-1, 1, m)).to(dev); y = torch.zeros(m, dtype=torch.float64, device=dev)
    for _ in range(5):
        assert pkg.dmv(pkg.OP_NONE, 1.0, A, d, x, 0.0, y) == 0
    torch.cuda.synchronize()
    t = np.fromfile(TRACE, dtype=np.uint64).reshape(-1, 8)
    rlen = np.diff(rp.astype(np.int64))
    st, tb, tl, ts, te = (t[:, k].astype(np.int64) for k in range(5))
    short = tl > 0                      # blocks that went through the LDS tile path (not a single long row)
    t0 = st.min()
    us = lambda a: a / 100.0
    print(json.dumps({"matrix": name, "m": m, "nnz": int(len(ci)), "row_blocks": int(len(t)), "long_row_blocks": int((~short).sum()),
                      "kernel_span_us (first start -> last end)": float(us(te.max() - t0)),
                      "start_us_q (dispatch ramp)": q(us(st - t0)), "end_us_q": q(us(te - t0)),
                      "block_table_us_q": q(us(tb - st)), "tile_to_lds_us_q": q(us(tl - tb)[short]),
                      "barrier_us_q": q(us(ts - tl)[short]), "reduce_rows_us_q": q(us(te - ts)[short]),
                      "whole_block_us_q": q(us(te - st)),
                      "long_row_block_us_q": q(us(te - tb)[~short]) if (~short).any() else None,
                      "rows_per_block_q": q((t[:, 5] >> np.uint64(32)).astype(np.int64)),
                      "quartiles": "[5, 25, 50, 75, 95, 100] %",
                      "start_us_median_by_block_index_mod_8 (= XCD)": [round(float(np.median(us(st - t0)[k::8])), 2) for k in range(8)],
                      "first_start_us_by_block_index_mod_8": [round(float(us(st - t0)[k::8].min()), 2) for k in range(8)],
                      "last_to_end (block index, start, table, tile, barrier, reduce, rows, nnz, longest row)":
                          [[int(b), float(us(st[b] - t0)), float(us(tb[b] - st[b])), float(us(tl[b] - tb[b])), float(us(ts[b] - tl[b])),
                            float(us(te[b] - max(ts[b], tb[b]))), int(t[b, 5] >> np.uint64(32)), int(t[b, 5] & np.uint64(0xffffffff)),
                            int(rlen[int(t[b, 6]):int(t[b, 6]) + int(t[b, 5] >> np.uint64(32))].max())]
                           for b in np.argsort(te)[-8:][::-1]]}))
