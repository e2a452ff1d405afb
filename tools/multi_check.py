#!/usr/bin/env python3
"""In-library multi-device csrmm (aoclsparse_mi355_dcsrmm_multi_slabs / _multi), ONE process over N visible GPUs.

  multi_check.py [--devices N] [--cols 256] [--grid 1000] [--layout row|col] [--reps 10] [--same-device]

Prints one JSON line: per-call wall time with N devices (every call returns when all devices are done: the hand-over to the slot
workers, the launches and the per-device stream synchronisation are inside it), the 1-device time of the same job, T1 / (N * TN), and whether
every device's slab equals the 1-device product bit for bit.  --same-device puts all N slots on device 0 (what a one-GPU
box can run: the control flow ONLY -- the line then says "same_device": true and carries NO efficiency figure, because slots that
share one GPU say nothing about scaling).  On distinct devices the line also carries each device's own time for its slab
(a one-device call of that slab's width on that device).  bench.py runs this as a CHILD process when it sees more than one GPU, with a
timeout, so that a problem on a multi-GPU node cannot take the bench line down with it.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--devices", type=int, default=0)
    ap.add_argument("--cols", type=int, default=256)
    ap.add_argument("--grid", type=int, default=1000)
    ap.add_argument("--layout", default="row", choices=["row", "col"])
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--same-device", action="store_true")
    args = ap.parse_args()
    pkg = entry.load_package()
    L = pkg.lib()
    ndev_visible = torch.cuda.device_count()
    n = args.devices or ndev_visible
    torch.cuda.set_device(0)
    st, dev0, cus, name = pkg.device_info()
    assert st == 0
    devices = [dev0] * n if args.same_device else [(dev0 + i) % ndev_visible for i in range(n)]
    m, rp, ci, v = entry.laplace5(args.grid)
    A = pkg.Matrix(0, m, m, rp, ci, v)
    d = pkg.Descr()
    assert L.aoclsparse_set_mm_hint(A.h, pkg.OP_NONE, d.h, 100) == 0 and L.aoclsparse_optimize(A.h) == 0
    colmaj = args.layout == "col"
    order = pkg.ORDER_COLUMN if colmaj else pkg.ORDER_ROW
    ncols = args.cols
    shards = [pkg.column_shard(ncols, n, r) for r in range(n)]
    gen = torch.Generator(device="cpu")
    gen.manual_seed(777)
    Bfull = torch.rand((ncols, m), dtype=torch.float64, generator=gen) * 2.0 - 1.0      # column j = row j of this tensor

    def slab(j0, j1, device):
        cols = Bfull[j0:j1]
        t = cols.contiguous() if colmaj else cols.t().contiguous()
        return t.reshape(-1).to(device)

    Bs, Cs = [], []
    for (j0, j1), dv in zip(shards, devices):
        device = torch.device("cuda", dv)
        Bs.append(slab(j0, j1, device))
        Cs.append(torch.zeros(max(j1 - j0, 1) * m, dtype=torch.float64, device=device))
    widths = [j1 - j0 for j0, j1 in shards]
    L.aoclsparse_mi355_set_pointer_mode(pkg.PTR_DEVICE)
    # per-slab leading dimensions differ for row-major slabs of unequal width: the entry point takes ONE ldb / ldc, so
    # row-major needs equal widths (ncols % (4 n) == 0); column-major slabs all have ld = m
    if not colmaj:
        assert len(set(widths)) == 1, "row-major slabs need ncols to be a multiple of 4 * devices"
    ldb = m if colmaj else widths[0]

    def multi():
        return pkg.dcsrmm_multi_slabs(pkg.OP_NONE, 1.0, A, d, order, Bs, ncols, ldb, 0.0, Cs, ldb, devices)

    t0 = time.perf_counter()
    st = multi()
    t_first = time.perf_counter() - t0   # includes building the replicas on the other devices
    assert st == 0, pkg.STATUS.get(st, st)
    for _ in range(2):
        assert multi() == 0
    laps = []
    for _ in range(args.reps):
        t0 = time.perf_counter()
        assert multi() == 0
        laps.append((time.perf_counter() - t0) * 1e3)
    # 1 device, same job (all columns), same entry point
    B1 = slab(0, ncols, torch.device("cuda", dev0))
    C1 = torch.zeros(ncols * m, dtype=torch.float64, device=torch.device("cuda", dev0))
    ld1 = m if colmaj else ncols
    one = lambda: pkg.dcsrmm_multi_slabs(pkg.OP_NONE, 1.0, A, d, order, [B1], ncols, ld1, 0.0, [C1], ld1, [dev0])
    for _ in range(3):
        assert one() == 0
    laps1 = []
    for _ in range(args.reps):
        t0 = time.perf_counter()
        assert one() == 0
        laps1.append((time.perf_counter() - t0) * 1e3)
    # parity: every slab against the matching columns of the 1-device product
    C1m = C1.reshape(ncols, m) if colmaj else C1.reshape(m, ncols)
    same = True
    for (j0, j1), c in zip(shards, Cs):
        if j1 <= j0:
            continue
        got = c[: (j1 - j0) * m].cpu()
        ref = (C1m[j0:j1] if colmaj else C1m[:, j0:j1]).contiguous().reshape(-1).cpu()
        same = same and bool(torch.equal(got, ref))
    tn, t1 = float(np.median(laps)), float(np.median(laps1))
    same_device = len(set(devices)) < len(devices)
    out = {"what": "aoclsparse_mi355_dcsrmm_multi_slabs, one process", "devices": devices, "same_device": same_device,
           "visible_gpus": ndev_visible, "device0": name, "layout": "column-major" if colmaj else "row-major", "ncols": ncols, "m": m,
           "cols_per_device": widths, "ms_wall_median": round(tn, 4), "ms_wall_min": round(min(laps), 4),
           "ms_one_device_wall_median": round(t1, 4),
           "first_call_ms_with_replica_build": round(t_first * 1e3, 1), "replicas": int(L.aoclsparse_mi355_replica_count(A.h)),
           "replicas_cloned_device_to_device": int(L.aoclsparse_mi355_replicas_cloned(A.h)),
           "slabs_bit_exact": same,
           "note": "wall clock around the call: the hand-over to the persistent slot workers, launches and the per-device "
                   "stream synchronisation are inside; beta = 0 with C read (default)"}
    if same_device:
        out["note"] += "; every slot shares ONE GPU: control flow only, no efficiency is reported"
    else:
        out["efficiency_wall"] = round(t1 / (n * tn), 4)
        import ctypes
        buf = (ctypes.c_float * 64)()
        assert multi() == 0
        cnt = L.aoclsparse_mi355_multi_last_ms(buf, 64)
        out["per_device"] = [{"device": dv, "cols": j1 - j0, "ms_wall": round(float(buf[i]), 4)}
                             for i, ((j0, j1), dv) in enumerate(zip(shards, devices)) if i < cnt]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
