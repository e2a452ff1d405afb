#!/usr/bin/env python3
"""round 5: the headline product back to back, one call per event pair, and COLD -- the 256 MB Infinity Cache flushed before every
product by a 1 GB fill (leaves dirty lines: their write-back runs into the product) or by a 1 GB read (clean).
python3 tools/exp_cold.py  ->  profiles/r5/sell_placement.txt"""
import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
import __graft_entry__ as entry
pkg = entry.load_package(); L = pkg.lib()
L.aoclsparse_mi355_set_pointer_mode(pkg.PTR_DEVICE)
m, rp, ci, v = entry.laplace5(4096)
A = pkg.Matrix(0, m, m, rp, ci, v); d = pkg.Descr()
assert L.aoclsparse_set_mv_hint(A.h, pkg.OP_NONE, d.h, 100) == 0 and L.aoclsparse_optimize(A.h) == 0
x = torch.from_numpy(np.sin(0.01 * np.arange(m))).cuda(); y = torch.zeros(m, dtype=torch.float64, device="cuda")
big = torch.ones(1 << 28, dtype=torch.float32, device="cuda")  # 1 GB
def cold(flush):
    t = []
    for _ in range(14):
        flush(); torch.cuda.synchronize(); pkg.timer_start()
        pkg.dmv(pkg.OP_NONE, 1.0, A, d, x, 0.0, y)
        t.append(pkg.timer_stop())
    return round(float(np.median(t[2:])), 4)
hot = []
for _ in range(3):
    torch.cuda.synchronize(); pkg.timer_start()
    for _ in range(50): pkg.dmv(pkg.OP_NONE, 1.0, A, d, x, 0.0, y)
    hot.append(round(pkg.timer_stop() / 50, 4))
one = []
for _ in range(14):
    torch.cuda.synchronize(); pkg.timer_start(); pkg.dmv(pkg.OP_NONE, 1.0, A, d, x, 0.0, y); one.append(pkg.timer_stop())
print({"hot_ms": hot, "single_call_event_pair_ms": round(float(np.median(one[2:])), 4),
       "cold_after_1GB_fill_ms": cold(lambda: big.fill_(1.0)), "cold_after_1GB_read_ms": cold(lambda: big.sum())})
