#!/usr/bin/env python3
"""Column-sharded csrmm self-check, started once per rank (torch.distributed.run or plain python for one rank):
every rank computes its slab of C = A*B through aocl-sparse_amd/sharded.py (A broadcast from rank 0, no data-path
collective), then ALSO the full single-rank product, and checks bit for bit that (i) its slab equals the matching
columns of the single-rank C and (ii) the all-gathered C equals the single-rank C.  Rank 0 prints one JSON line.
--backend gloo lets the ranks share one GPU (the 1-GPU test box); nccl is the driver's multi-GPU configuration."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="gloo", choices=["gloo", "nccl"])
    ap.add_argument("--grid", type=int, default=200)
    ap.add_argument("--cols", type=int, default=40)
    ap.add_argument("--layout", default="col", choices=["col", "row"])
    ap.add_argument("--beta", type=float, default=0.0)
    args = ap.parse_args()
    import numpy as np
    import torch
    import torch.distributed as dist

    import __graft_entry__ as entry

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "the product has no CPU path"
    di = local % torch.cuda.device_count() if args.backend == "gloo" else local
    torch.cuda.set_device(di)
    device = torch.device("cuda", di)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    D = dist if world > 1 else None
    pkg = entry.load_package()
    import aocl_sparse_amd.sharded as sharded

    pkg.lib().aoclsparse_mi355_set_pointer_mode(pkg.PTR_DEVICE)
    csr = None
    if rank == 0:
        m, rp, ci, v = entry.laplace5(args.grid)
        v = v * np.random.default_rng(11).uniform(0.5, 1.5, len(v))  # inexact products
        csr = (m, m, rp, ci, v)
    sh = sharded.ShardedCsrmm(pkg, torch, D, device, rank, world, csr, args.cols, args.layout)
    m = sh.m
    B = sh.make_B()
    gen = torch.Generator(device=device)
    gen.manual_seed(5)
    C0full = torch.rand(args.cols * m, dtype=torch.float64, device=device, generator=gen)  # as (cols, m) columns
    cols0 = C0full.reshape(args.cols, m)
    mine = cols0[sh.j0:sh.j1]
    C = (mine if args.layout == "col" else mine.t()).contiguous().reshape(-1).clone()
    assert sh.run(B, C, alpha=1.5, beta=args.beta) == 0
    torch.cuda.synchronize()
    gathered, ag_ms = sh.gather_C(C)
    # the same job on ONE rank: all columns through the same handle
    Bf = sh.make_B(j0=0, j1=args.cols)
    Cf = (cols0 if args.layout == "col" else cols0.t()).contiguous().reshape(-1).clone()
    assert sh.run(Bf, Cf, alpha=1.5, beta=args.beta, nloc=args.cols) == 0
    torch.cuda.synchronize()
    full_cols = Cf.reshape(args.cols, m) if args.layout == "col" else Cf.reshape(m, args.cols).t().contiguous()
    my_cols = C.reshape(sh.nloc, m) if args.layout == "col" else C.reshape(m, sh.nloc).t().contiguous()
    slab_ok = bool(torch.equal(my_cols, full_cols[sh.j0:sh.j1]))
    gather_ok = bool(torch.equal(gathered, full_cols))
    # the in-library shard entry on the FULL arrays must write exactly the same slab
    Cs = (cols0 if args.layout == "col" else cols0.t()).contiguous().reshape(-1).clone()
    order = pkg.ORDER_COLUMN if args.layout == "col" else pkg.ORDER_ROW
    ld = m if args.layout == "col" else args.cols
    st = pkg.dcsrmm_shard(pkg.OP_NONE, 1.5, sh.A, sh.descr, order, Bf, args.cols, sh.n if args.layout == "col" else args.cols,
                          args.beta, Cs, ld, world, rank)
    torch.cuda.synchronize()
    cs_cols = Cs.reshape(args.cols, m) if args.layout == "col" else Cs.reshape(m, args.cols).t().contiguous()
    untouched = torch.ones(args.cols, dtype=torch.bool, device=device)
    untouched[sh.j0:sh.j1] = False
    abi_ok = bool(st == 0 and torch.equal(cs_cols[sh.j0:sh.j1], full_cols[sh.j0:sh.j1])
                  and torch.equal(cs_cols[untouched], cols0[untouched]))
    ok_all = sharded.reduce_scalar(1.0 if (slab_ok and gather_ok and abi_ok) else 0.0, "min", D, device)
    if rank == 0:
        print(json.dumps({"world": world, "backend": args.backend if world > 1 else "none", "layout": args.layout,
                          "cols": args.cols, "m": m, "shard": [sh.j0, sh.j1], "slab_bit_exact": slab_ok,
                          "gathered_bit_exact": gather_ok, "abi_shard_bit_exact": abi_ok, "all_ranks_ok": ok_all == 1.0,
                          "a_broadcast_ms": round(sh.a_broadcast_ms, 3), "c_allgather_ms": round(ag_ms, 3)}))
    if world > 1:
        dist.destroy_process_group()
    sys.exit(0 if ok_all == 1.0 else 1)


if __name__ == "__main__":
    main()
