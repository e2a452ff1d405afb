#!/usr/bin/env python3
"""round 5 experiment: the same SELL-64 handle built several times in one process -- does the time of aoclsparse_dmv depend on WHEN
(i.e. where) its arrays were allocated?  (shell-like ran at 90 us in some bench runs and 102 us in others with the same binary.)"""
import os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # (tools/history/ -> repository root)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as entry, standins
pkg = entry.load_package(); L = pkg.lib()
L.aoclsparse_mi355_set_pointer_mode(pkg.PTR_DEVICE)
name = sys.argv[1] if len(sys.argv) > 1 else "shell-like"
label, m, rp, ci, v = standins.load(name)
d = pkg.Descr()
hs = []
pre = os.environ.get("PRE_ALLOC_MB")
junk = torch.empty(int(pre) << 20, dtype=torch.uint8, device="cuda") if pre else None
for k in range(4):
    A = pkg.Matrix(0, m, m, rp, ci, v)
    assert L.aoclsparse_set_mv_hint(A.h, pkg.OP_NONE, d.h, 1000) == 0 and L.aoclsparse_optimize(A.h) == 0
    hs.append(A)
x = torch.from_numpy(np.random.default_rng(1).uniform(-1, 1, m)).cuda(); y = torch.zeros(m, dtype=torch.float64, device="cuda")
print("x %x y %x" % (x.data_ptr(), y.data_ptr()), file=sys.stderr)
res = [[] for _ in hs]
for rep in range(4):
    for k, A in enumerate(hs):
        for _ in range(20): pkg.dmv(pkg.OP_NONE, 1.0, A, d, x, 0.0, y)
        torch.cuda.synchronize(); pkg.timer_start()
        for _ in range(200): pkg.dmv(pkg.OP_NONE, 1.0, A, d, x, 0.0, y)
        res[k].append(round(pkg.timer_stop() / 200 * 1e3, 2))
print(json.dumps({"matrix": label, "pre_alloc_mb": pre, "us_per_handle_in_creation_order": res}), flush=True)
# the same handles with the Infinity Cache flushed before every product (a 1 GB fill between two timed calls): does the spread
# between placements survive when nothing of the matrix can be left in the cache from the call before?
big = torch.ones(1 << 28, dtype=torch.float32, device="cuda")
for how, flush in (("1 GB fill (dirty lines: their write-back runs into the product)", lambda: big.fill_(1.0)), ("1 GB read", lambda: big.sum())):
    cold = []
    for k, A in enumerate(hs):
        t = []
        for _ in range(30):
            flush()
            torch.cuda.synchronize(); pkg.timer_start()
            pkg.dmv(pkg.OP_NONE, 1.0, A, d, x, 0.0, y)
            t.append(pkg.timer_stop() * 1e3)
        cold.append(round(float(np.median(t)), 2))
    print(json.dumps({"matrix": label, "flush": how, "us_per_handle_cold_median_of_30": cold}), flush=True)
