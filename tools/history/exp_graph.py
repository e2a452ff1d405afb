#!/usr/bin/env python3
"""Can the library's device-pointer calls be captured into a HIP graph (torch.cuda.CUDAGraph on the stream handed to
aoclsparse_mi355_set_stream)?  Short kernels are launch-bound: 20 SpMVs / one L+U solve pair per replay.  Diagnostic."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # (tools/history/ -> repository root)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as entry, standins
pkg = entry.load_package(); L = pkg.lib()
L.aoclsparse_mi355_set_pointer_mode(pkg.PTR_DEVICE)
dev = torch.device("cuda", 0)
import ctypes
def run(name, gen, reps=20):
    m, rp, ci, v = gen()
    A = pkg.Matrix(0, m, m, rp, ci, v); d = pkg.Descr()
    assert L.aoclsparse_set_mv_hint(A.h, pkg.OP_NONE, d.h, 1000) == 0 and L.aoclsparse_optimize(A.h) == 0
    x = torch.from_numpy(np.random.default_rng(1).uniform(-1, 1, m)).to(dev)
    y = torch.zeros(m, dtype=torch.float64, device=dev)
    s = torch.cuda.Stream()
    assert L.aoclsparse_mi355_set_stream(ctypes.c_void_p(s.cuda_stream)) == 0
    with torch.cuda.stream(s):
        for _ in range(3):
            assert pkg.dmv(pkg.OP_NONE, 1.0, A, d, x, 0.0, y) == 0
        s.synchronize()
        yref = y.clone()
        # eager, back to back
        t0 = time.perf_counter()
        for _ in range(50 * reps):
            pkg.dmv(pkg.OP_NONE, 1.0, A, d, x, 0.0, y)
        s.synchronize()
        eager = (time.perf_counter() - t0) / (50 * reps)
        g = torch.cuda.CUDAGraph()
        y.zero_()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                st = pkg.dmv(pkg.OP_NONE, 1.0, A, d, x, 0.0, y)
        assert st == 0
        g.replay(); s.synchronize()
        same = bool(torch.equal(y, yref))
        t0 = time.perf_counter()
        for _ in range(50):
            g.replay()
        s.synchronize()
        graph = (time.perf_counter() - t0) / (50 * reps)
    print(json.dumps({"what": "aoclsparse_dmv x%d per graph replay" % reps, "matrix": name, "eager_us_per_call": round(eager * 1e6, 2),
                      "graph_us_per_call": round(graph * 1e6, 2), "bit_identical": same}), flush=True)
run("circuit-like", standins.circuit_like)
run("web-like", standins.web_like)
L.aoclsparse_mi355_set_stream(None)
