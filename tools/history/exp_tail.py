#!/usr/bin/env python3
"""Experiment: how much of the web-like time is the long-row tail?  Same generator, row lengths clipped."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import __graft_entry__ as entry
import standins
pkg = entry.load_package(); L = pkg.lib(); dev = torch.device("cuda", 0)
L.aoclsparse_mi355_set_pointer_mode(pkg.PTR_DEVICE)
d0 = pkg.Descr()
m, rp, ci, v = standins.ALL["web-like"]()
for clip in (10**9, 512, 128, 32):
    lens = np.minimum(np.diff(rp), clip)
    keep = np.concatenate([np.arange(rp[i], rp[i] + lens[i]) for i in np.flatnonzero(np.diff(rp) > clip)]) if clip < 10**9 else None
    if clip < 10**9:
        mask = np.ones(len(ci), bool)
        for i in np.flatnonzero(np.diff(rp) > clip):
            mask[rp[i] + clip:rp[i + 1]] = False
        ci2, v2 = ci[mask], v[mask]
        rp2 = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    else:
        rp2, ci2, v2 = rp, ci, v
    x = torch.from_numpy(np.random.default_rng(1).uniform(-1, 1, m)).to(dev); y = torch.zeros(m, dtype=torch.float64, device=dev)
    A = pkg.Matrix(0, m, m, rp2, ci2, v2)
    assert L.aoclsparse_set_mv_hint(A.h, pkg.OP_NONE, d0.h, 100) == 0 and L.aoclsparse_optimize(A.h) == 0
    for _ in range(5): pkg.dmv(pkg.OP_NONE, 1.0, A, d0, x, 0.0, y)
    torch.cuda.synchronize(); pkg.timer_start()
    for _ in range(50): pkg.dmv(pkg.OP_NONE, 1.0, A, d0, x, 0.0, y)
    ms = pkg.timer_stop() / 50
    print(json.dumps(dict(clip=clip, nnz=len(v2), kernel=A.spmv_info().kernel, blocks=A.spmv_info().row_blocks, us=round(ms * 1e3, 2))), flush=True)
