"""CPU tier: the multi-rank bookkeeping of bench.py (column shards, max-over-ranks time, whole-job
throughput) under torch.distributed with the gloo backend, world_size 2 -- the N > 1 path the driver
runs on 2/4/8 GPUs with RCCL."""
import os
import socket
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def test_column_shards_partition():
    for ncols in (256, 255, 7, 1):
        for world in (1, 2, 3, 4, 8):
            cover = []
            for r in range(world):
                j0, j1 = bench.column_shard(ncols, world, r)
                assert 0 <= j0 <= j1 <= ncols
                cover += list(range(j0, j1))
            assert cover == list(range(ncols))
            sizes = [bench.column_shard(ncols, world, r)[1] - bench.column_shard(ncols, world, r)[0]
                     for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
    assert bench.column_shard(256, 8, 3) == (96, 128)


def test_byte_models():
    # tests/include/aoclsparse_gbyte.hpp:39-45 on the L100 matrix: 795,204 B
    assert bench.spmv_bytes(10000, 10000, 49600) == 795204
    assert bench.spmv_bytes(10000, 10000, 49600, True) == 795204 + 80000
    m, nnz = 1000000, 4996000
    assert bench.csrmm_bytes(m, m, nnz, 256) == (m + 1 + nnz) * 4 + nnz * 8 + 8 * 256 * 2 * m


def _worker(rank, world, port, q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # rank r "processed" (r+1)*10 units in (r+1) seconds
        thr, tmax = bench.job_throughput((rank + 1) * 10.0, float(rank + 1), dist)
        s = bench.reduce_scalar(rank + 1.0, "sum", dist)
        j0, j1 = bench.column_shard(256, world, rank)
        cols = bench.reduce_scalar(j1 - j0, "sum", dist)
        q.put((rank, thr, tmax, s, cols))
    finally:
        dist.destroy_process_group()


def test_job_throughput_world2_gloo():
    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, thr, tmax, ssum, cols in res:
        assert tmax == 2.0 and thr == 30.0 / 2.0 and ssum == 3.0 and cols == 256.0


def test_single_rank_is_identity():
    assert bench.reduce_scalar(3.5, "max") == 3.5
    assert bench.job_throughput(10.0, 2.0) == (5.0, 2.0)


def test_matrix_market_reader(tmp_path):
    """tools/standins.read_mtx: coordinate real symmetric -> full sorted CSR (what the reference's harness does,
    tests/include/aoclsparse_init.hpp:452-694)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import standins
    p = tmp_path / "t.mtx"
    p.write_text("%%MatrixMarket matrix coordinate real symmetric\n% c\n4 4 5\n1 1 2.0\n2 1 -1.0\n3 3 4.0\n4 2 0.5\n4 4 1.0\n")
    m, n, rp, ci, v = standins.read_mtx(str(p))
    assert (m, n) == (4, 4) and list(rp) == [0, 2, 4, 5, 7]
    assert list(ci) == [0, 1, 0, 3, 2, 1, 3] and list(v) == [2.0, -1.0, -1.0, 0.5, 4.0, 0.5, 1.0]
    q = tmp_path / "p.mtx"
    q.write_text("%%MatrixMarket matrix coordinate pattern general\n2 3 2\n1 3\n2 1\n")
    m, n, rp, ci, v = standins.read_mtx(str(q))
    assert (m, n, list(rp), list(ci)) == (2, 3, [0, 1, 2], [2, 0]) and len(v) == 2
