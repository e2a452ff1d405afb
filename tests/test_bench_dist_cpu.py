"""CPU tier: the multi-rank host logic of bench.py and aocl-sparse_amd/sharded.py (column shards, CSR broadcast,
slab generation, C all-gather, max-over-ranks time, whole-job throughput) under torch.distributed with the gloo
backend, world_size 2 -- the N > 1 path the driver runs on 2/4/8 GPUs with RCCL.  No compute: the product has no
CPU path (the csrmm itself is covered by the 2-process test of tests/test_gpu_parity.py on the GPU)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import __graft_entry__ as entry  # noqa: E402
import bench  # noqa: E402


def test_column_shards_partition():
    """The reference's thread split (csrmm_kt.cpp:68-82): start = n*t/T rounded up to a multiple of 4, capped at n.
    The python rule of bench.py and the library's aoclsparse_mi355_column_shard must agree."""
    pkg = entry.load_package()
    for ncols in (256, 255, 64, 7, 4, 1, 0):
        for world in (1, 2, 3, 4, 8):
            cover = []
            for r in range(world):
                j0, j1 = bench.column_shard(ncols, world, r)
                assert (j0, j1) == pkg.column_shard(ncols, world, r)
                assert 0 <= j0 <= j1 <= ncols
                assert j0 % 4 == 0 or j0 == ncols  # slabs start on the reference's 4-column blocks
                cover += list(range(j0, j1))
            assert cover == list(range(ncols))
    assert bench.column_shard(256, 8, 3) == (96, 128)
    assert [bench.column_shard(7, 2, r) for r in range(2)] == [(0, 4), (4, 7)]
    assert [bench.column_shard(10, 4, r) for r in range(4)] == [(0, 4), (4, 8), (8, 8), (8, 10)]
    with pytest.raises(ValueError):
        pkg.column_shard(8, 2, 2)


def test_byte_models():
    # tests/include/aoclsparse_gbyte.hpp:39-45 on the L100 matrix: 795,204 B
    assert bench.spmv_bytes(10000, 10000, 49600) == 795204
    assert bench.spmv_bytes(10000, 10000, 49600, True) == 795204 + 80000
    m, nnz = 1000000, 4996000
    assert bench.csrmm_bytes(m, m, nnz, 256) == (m + 1 + nnz) * 4 + nnz * 8 + 8 * 256 * 2 * m
    assert bench.trsv_bytes(7, 10) == (7 + 1 + 10) * 4 + (14 + 10) * 8


def test_quartiles_and_roofline():
    q = bench.quartiles([5.0, 1.0, 3.0, 2.0, 4.0])
    assert (q["min"], q["q1"], q["median"], q["q3"], q["max"], q["n"]) == (1.0, 2.0, 3.0, 4.0, 5.0, 5)
    r = bench.roofline(8e9, 1.0)  # 8 GB in 1 ms = 8 TB/s
    assert r["achieved"] == 8000.0 and r["frac"] == 1.0 and r["bound"] == "hbm" and r["traffic"] is None
    model, logical, phys = bench.cpu_info()
    assert logical >= 1 and 1 <= phys <= max(logical, phys)


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pkg = entry.load_package()
        import aocl_sparse_amd.sharded as sharded

        # rank r "processed" (r+1)*10 units in (r+1) seconds
        thr, tmax = bench.job_throughput((rank + 1) * 10.0, float(rank + 1), dist)
        s = bench.reduce_scalar(rank + 1.0, "sum", dist)
        mn = sharded.reduce_scalar(rank + 1.0, "min", dist)
        # A is broadcast from rank 0: every rank ends up with the same five arrays
        csr = None
        if rank == 0:
            m, rp, ci, v = entry.laplace5(12)
            csr = (m, m, rp, ci, v)
        (m, n, rp, ci, v), ms = sharded.broadcast_csr(dist, torch, "cpu", rank, csr)
        m0, rp0, ci0, v0 = entry.laplace5(12)
        same = m == m0 and n == m0 and np.array_equal(rp, rp0) and np.array_equal(ci, ci0) and np.array_equal(v, v0)
        # slabs: seeded per column, so the concatenation of the ranks' slabs is the 1-rank matrix
        ncols = 10
        shards = [pkg.column_shard(ncols, world, r) for r in range(world)]
        j0, j1 = shards[rank]
        Bc = sharded.make_B_slab(torch, "cpu", m, j0, j1, "col")
        Br = sharded.make_B_slab(torch, "cpu", m, j0, j1, "row")
        full = sharded.make_B_slab(torch, "cpu", m, 0, ncols, "col").reshape(ncols, m)
        slab_ok = torch.equal(Bc.reshape(j1 - j0, m), full[j0:j1]) and torch.equal(Br.reshape(m, j1 - j0).t(), full[j0:j1])
        gathered, _ = sharded.gather_slabs(torch, dist, "cpu", rank, shards, Bc.reshape(j1 - j0, m))
        # more ranks than 4-column blocks: 4 columns over 2 ranks -> the second rank owns nothing; its slab is a
        # (0, m) tensor, the gather still returns the whole matrix everywhere
        shards4 = [pkg.column_shard(4, world, r) for r in range(world)]
        a0, a1 = shards4[rank]
        slab4 = sharded.make_B_slab(torch, "cpu", m, a0, a1, "col").reshape(a1 - a0, m)
        g4, _ = sharded.gather_slabs(torch, dist, "cpu", rank, shards4, slab4)
        empty_ok = shards4 == [(0, 4), (4, 4)] and torch.equal(g4, full[:4])
        # row-sharded SpMV (SURVEY 8e "next"): 13 rows over 2 ranks -> 6 + 7, the slices' all-gather rebuilds the vector
        mrow = 13
        a, b = sharded.row_shard(mrow, world, rank)
        yloc = torch.arange(a, b, dtype=torch.float64) * 1.5
        xfull = torch.full((mrow,), -1.0, dtype=torch.float64)
        gms = sharded.allgather_rows(torch, dist, "cpu", rank, world, mrow, yloc, xfull)
        empty_ok = empty_ok and gms >= 0.0 and torch.equal(xfull, torch.arange(mrow, dtype=torch.float64) * 1.5)
        info = sharded.communicator_info(dist, torch)
        empty_ok = empty_ok and info == {"backend": "gloo", "world": world}
        # row-sharded sp2m: handles over the rank's rows of A and the whole of B are host objects (no product here:
        # that needs the GPU); the slices' sizes are all-gathered into the offset of the rank's rows in the global C
        g = 9
        mB, rpB, ciB, vB = entry.laplace5(g)
        s2 = sharded.ShardedSp2m(pkg, torch, dist, "cpu", rank, world, g * g, lambda r0, r1: entry.laplace5_rows(g, r0, r1),
                                 (mB, mB, rpB, ciB, vB))
        first, total = s2.slice_offsets(100 + 11 * rank)
        empty_ok = empty_ok and (s2.r0, s2.r1) == sharded.row_shard(g * g, world, rank) and total == sum(100 + 11 * r for r in range(world))
        empty_ok = empty_ok and first == sum(100 + 11 * r for r in range(rank)) and s2.nnz_a_loc == int(rpB[s2.r1] - rpB[s2.r0])
        q.put((rank, thr, tmax, s, mn, bool(same), bool(slab_ok), bool(torch.equal(gathered, full)) and bool(empty_ok), shards))
    finally:
        dist.destroy_process_group()


def test_sharded_host_logic_world2_gloo():
    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, thr, tmax, ssum, mn, same, slab_ok, gather_ok, shards in res:
        assert tmax == 2.0 and thr == 30.0 / 2.0 and ssum == 3.0 and mn == 1.0
        assert same, "broadcast CSR differs from rank 0's"
        assert slab_ok and gather_ok
        assert shards == [(0, 8), (8, 10)]  # 10 columns over 2 ranks: 4-column blocks, remainder to the last


def test_row_shards_and_slices():
    """row_shard partitions the rows contiguously (balanced to one row); slice_rows keeps the index base and rebases the row
    pointers; the slices together hold every entry once; SpMV on the slices equals the rows of the full product."""
    pkg = entry.load_package()
    import aocl_sparse_amd.sharded as sharded
    import oracle
    for base in (0, 1):
        m, rp, ci, v = entry.laplace5(9, base=base)
        x = np.cos(0.3 * np.arange(m))
        so, yfull = oracle.dcsrmv(0, base, 1.0, m, len(v), v, ci, rp, x, 0.0, np.zeros(m))
        for world in (1, 2, 3, 7, 100):
            rows = [sharded.row_shard(m, world, r) for r in range(world)]
            assert rows[0][0] == 0 and rows[-1][1] == m and all(rows[i][1] == rows[i + 1][0] for i in range(world - 1))
            assert max(b - a for a, b in rows) - min(b - a for a, b in rows) <= 1
            total = 0
            for a, b in rows:
                ml, nl, rpl, cil, vl = sharded.slice_rows((m, m, rp, ci, v), a, b)
                assert ml == b - a and nl == m and rpl[0] == base and rpl[-1] - base == len(vl) == len(cil)
                total += len(vl)
                if ml:
                    so, yl = oracle.dcsrmv(0, base, 1.0, ml, len(vl), vl, cil, rpl, x, 0.0, np.zeros(ml))
                    assert np.array_equal(yl, yfull[a:b])
            assert total == len(v)


def test_single_rank_is_identity():
    assert bench.reduce_scalar(3.5, "max") == 3.5
    assert bench.job_throughput(10.0, 2.0) == (5.0, 2.0)
    import torch
    pkg = entry.load_package()
    import aocl_sparse_amd.sharded as sharded
    m, rp, ci, v = entry.laplace5(5)
    csr, ms = sharded.broadcast_csr(None, torch, "cpu", 0, (m, m, rp, ci, v))
    assert ms == 0.0 and csr[0] == m
    cols = torch.arange(12, dtype=torch.float64).reshape(3, 4)
    out, ms = sharded.gather_slabs(torch, None, "cpu", 0, [(0, 3)], cols)
    assert out is cols and ms == 0.0
    assert sharded.quartiles([]) is None


def test_matrix_market_reader(tmp_path):
    """tools/standins.read_mtx: coordinate real symmetric -> full sorted CSR (what the reference's harness does,
    tests/include/aoclsparse_init.hpp:452-694)."""
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


def test_block_standins_are_sorted_duplicate_free():
    """The block stand-ins are built directly in sorted order (tools/standins._block_csr): rows ascending and
    duplicate-free, full diagonal, block structure as documented."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import standins
    for gen, kw, bs in ((standins.shell_like, dict(n=5 * 37 * 9, width=37), 5), (standins.flan_like, dict(nx=5, ny=4, nz=6), 3)):
        m, rp, ci, v = gen(**kw)
        assert rp[0] == 0 and rp[-1] == len(ci) == len(v) and m % bs == 0
        for i in range(m):
            c = ci[rp[i]:rp[i + 1]]
            assert np.all(np.diff(c) > 0) and i in c
            assert 4.0 <= v[rp[i] + int(np.searchsorted(c, i))] <= 8.0
        # the bs rows of a node share one column pattern (what the csrmm row groups detect)
        for i in range(0, m, bs):
            for a in range(1, bs):
                assert np.array_equal(ci[rp[i]:rp[i + 1]], ci[rp[i + a]:rp[i + a + 1]])


def test_compact_record_fits_the_drivers_tail():
    """Round 3's 29 KB single line no longer fitted the driver's bounded tail of stdout (BENCH_r03.parsed = null).  The short
    record built from that very report must be strict JSON of at most 4 KB and carry roofline, cpu_baseline, parity and one
    number per leg; parsing ONLY the final 8 KB of an output that ends with it must yield them."""
    import json

    with open(os.path.join(ROOT, "profiles", "r3", "bench_default.json")) as f:
        full = json.loads([ln for ln in f.read().splitlines() if ln.startswith("{")][-1])
    assert len(json.dumps(full)) > 20000
    line = bench.compact_record(full, "bench_legs.json")
    assert len(line) <= bench.COMPACT_LIMIT and "\n" not in line
    json.loads(line, parse_constant=lambda c: (_ for _ in ()).throw(ValueError("not strict JSON: " + c)))
    stdout = "x" * 50000 + "\n" + json.dumps(full) + "\n" + line + "\n"
    tail = stdout[-8192:]
    rec = json.loads(tail[tail.rstrip("\n").rfind("\n") + 1:])
    assert rec["metric"].startswith("CSR SpMV fp64") and rec["unit"] == "GFLOP/s" and rec["value"] == full["value"]
    assert rec["roofline"]["frac"] == full["roofline"]["frac"] and rec["roofline"]["kernel_ms"] == full["roofline"]["kernel_ms"]
    assert rec["roofline"]["traffic"] == full["roofline"]["traffic"] and rec["roofline"]["peak"] == 8000.0
    # the real-traffic twin of frac (round 5): PMC bytes per launch / hipEvent time / peak
    assert abs(rec["roofline"]["frac_traffic"] - full["roofline"]["traffic"] / (full["roofline"]["kernel_ms"] * 1e-3) / 8e12) < 1e-3
    assert rec["roofline"]["frac_traffic"] < rec["roofline"]["frac"]
    cb = rec["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] == full["cpu_baseline"]["value"] and cb["cores"] == full["cpu_baseline"]["cores"]
    assert rec["parity"]["bit_exact"] is True and rec["config"]["communicator"] == full["config"]["communicator"]
    n = rec["legs"]
    for k in ("l100_us", "csr_adaptive_frac", "mix_frac_mean", "csrmm_row_ms", "csrmm_row_slab_ms", "csrmm_row_eff8", "csrmm_col_ms",
              "csrmm_col_slab_ms", "csrmm_col_eff8", "trsv_ms"):
        assert isinstance(n[k], float) and n[k] > 0, k
    # eff8 = T1 / (8 T_slab) from the two driver-visible numbers
    assert n["csrmm_row_eff8"] == round(n["csrmm_row_ms"] / (8 * n["csrmm_row_slab_ms"]), 4)
    # NaN / Inf in a leg must not break strictness; an absurdly long description must not break the size
    bad = dict(full, parity={"bit_exact": False, "max_abs_diff": float("nan")})
    bad["config"] = dict(full["config"], workload="w" * 5000, kernel="k" * 5000)
    line2 = bench.compact_record(bench._sanitize(bad))
    assert len(line2) <= bench.COMPACT_LIMIT and json.loads(line2)["parity"]["max_abs_diff"] is None
    # round 5's full report: the launch-bound configs[1] literal carries the C caller's per-call cost and the in-graph time next
    # to the event-per-call mean, and the record still fits
    with open(os.path.join(ROOT, "profiles", "r5", "bench_legs_default.json")) as f:
        full5 = json.load(f)
    line5 = bench.compact_record(full5, "bench_legs.json")
    assert len(line5) <= bench.COMPACT_LIMIT
    n5 = json.loads(line5)["legs"]
    assert n5["l100_c_caller_us"] == full5["l100"]["c_caller"]["dmv_pointer_mode_device"]["total_us"] < n5["l100_us"]
    assert n5["l100_graph_us"] == full5["l100"]["us_per_call_in_a_hip_graph_of_100"] < n5["l100_c_caller_us"]
    assert n5["l100_c_caller_us"] > 0
    # a run without legs (N > 1, --legs none)
    bare = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "config", "roofline")}
    bare["cpu_baseline"] = None
    assert json.loads(bench.compact_record(bare))["legs"] == {}


def test_bench_gpus_n_starts_its_own_ranks_dry():
    """`python bench.py --gpus 2` with no torch.distributed.run around it (the driver's command form in BENCH_r03.json.cmd): the
    parent starts both ranks as a child process, relays rank 0's line as its last line and returns the child's code.  --dry-launch
    stops after the rendezvous (no GPU here); a failing child (no GPU: the product has no CPU path) must give a non-zero code
    and no record; a WRONG WORLD_SIZE from an outer launcher stays a hard failure."""
    import json
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"]
    r = subprocess.run(cmd + ["--dry-launch"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    rec = json.loads(r.stdout.splitlines()[-1])
    assert rec == {"metric": "launch-check", "value": None, "n_gpus": 2, "rank_sum": 1.0, "self_launched": True}
    r = subprocess.run(cmd + ["--dry-launch"], cwd=ROOT, env=dict(env, WORLD_SIZE="3"), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run(cmd + ["--backend", "gloo", "--steps", "1", "--record", ""], cwd=ROOT, env=env, capture_output=True,
                           text=True, timeout=300)
        assert r.returncode != 0 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert "needs a GPU" in r.stderr


def test_laplace5_rows_are_the_rows_of_laplace5():
    g = 7
    m, rp, ci, v = entry.laplace5(g)
    for r0, r1 in ((0, 49), (0, 20), (20, 49), (13, 14), (5, 5)):
        ml, n, rpl, cil, vl = entry.laplace5_rows(g, r0, r1)
        assert ml == r1 - r0 and n == m and rpl[0] == 0
        assert np.array_equal(rpl, rp[r0:r1 + 1] - rp[r0]) and np.array_equal(cil, ci[rp[r0]:rp[r1]]) and np.array_equal(vl, v[rp[r0]:rp[r1]])


def test_library_communicator_control_flow_with_mocks():
    """The "nccl" path of sharded.broadcast_handle (library RCCL communicator: id drawn on every rank as a loadability probe,
    rank 0's id broadcast, collective init, broadcast_matrix, agreed fallback) cannot run with two ranks on a one-GPU box (RCCL
    refuses two ranks on one GPU) and never runs on CPU.  Its CONTROL FLOW is executed here against mocks of torch.distributed
    and of the library, once as rank 0 and once as rank 1, including the fallbacks: a rank that cannot load librccl must send
    everybody down the torch.distributed path BEFORE anyone enters the blocking ncclCommInitRank."""
    import ctypes
    import torch

    pkg = entry.load_package()
    import aocl_sparse_amd.sharded as sharded

    class FakeDist:
        class ReduceOp:
            MAX, SUM, MIN = "max", "sum", "min"

        def __init__(self, others_ok=True):
            self.others_ok, self.log = others_ok, []

        def is_initialized(self):
            return True

        def get_backend(self):
            return "nccl"

        def get_world_size(self):
            return 2

        def barrier(self):
            self.log.append("barrier")

        def broadcast(self, t, src):
            self.log.append(("broadcast", tuple(t.shape), src))

        def all_reduce(self, t, op=None):
            self.log.append(("all_reduce", op))
            if op == "min" and not self.others_ok:
                t.fill_(0.0)  # some other rank reported a failure

    class FakeLib:
        def __init__(self, load_ok=True):
            self.load_ok, self.calls = load_ok, []

        def aoclsparse_mi355_comm_unique_id(self, cid):
            self.calls.append("unique_id")
            return 0 if self.load_ok else 1

        def aoclsparse_mi355_comm_init(self, world, rank, cid):
            self.calls.append(("init", world, rank))
            return 0

        def aoclsparse_mi355_comm_broadcast_matrix(self, href, root):
            self.calls.append(("broadcast_matrix", root))
            ctypes.cast(href, ctypes.POINTER(ctypes.c_void_p))[0] = ctypes.c_void_p(0x1234)
            return 0

        def aoclsparse_mi355_synchronize(self):
            return 0

    class FakeMatrix:
        def __init__(self, h):
            self.h = h

        @classmethod
        def from_handle(cls, h, double=True):
            return cls(h)

    class FakePkg:
        CommId, MM_STATE_BUFFERS, STATUS, MmState = pkg.CommId, pkg.MM_STATE_BUFFERS, pkg.STATUS, pkg.MmState
        Matrix = FakeMatrix

        def __init__(self, lib):
            self._lib = lib

        def lib(self):
            return self._lib

    for rank in (0, 1):
        sharded._LIB_COMM.update(tried=False, ok=False)
        lib, dist = FakeLib(), FakeDist()
        A = FakeMatrix(ctypes.c_void_p(0x1234)) if rank == 0 else None
        out, ms, how = sharded.broadcast_handle(FakePkg(lib), torch, dist, "cpu", rank, 2, A)
        assert "library RCCL communicator" in how and out.h.value == 0x1234 and ms >= 0
        assert lib.calls == ["unique_id", ("init", 2, rank), ("broadcast_matrix", 0)]
        assert ("broadcast", (128,), 0) in dist.log
        assert sharded._LIB_COMM == {"tried": True, "ok": True}
        # the communicator is set up once per process
        lib.calls.clear()
        sharded.broadcast_handle(FakePkg(lib), torch, dist, "cpu", rank, 2, A)
        assert lib.calls == [("broadcast_matrix", 0)]
    # a rank that cannot load librccl: nobody calls comm_init (it would block the others forever)
    for load_ok, others_ok in ((False, True), (True, False)):
        sharded._LIB_COMM.update(tried=False, ok=False)
        lib, dist = FakeLib(load_ok), FakeDist(others_ok)
        assert sharded.library_communicator(FakePkg(lib), torch, dist, "cpu", 1, 2) is False
        assert lib.calls == ["unique_id"] and sharded._LIB_COMM == {"tried": True, "ok": False}
    sharded._LIB_COMM.update(tried=False, ok=False)


def test_sharded_csrmm_world8_column_ranges_and_state_bytes_with_mocks():
    """Round 6, multi-GPU readiness without hardware.  (a) ShardedCsrmm on a mock world of 8: the eight column ranges for n = 256
    and for the ragged n = 250 are the reference's thread split (level3/aoclsparse_csrmm_kt.cpp:68-82: start = n t / T rounded up
    to a multiple of 4 and capped at n), they tile [0, n) and every rank multiplies exactly its own slab.  (b) the
    torch.distributed wire of broadcast_handle: what rank 0 exports with aoclsparse_mi355_mm_state_export reaches
    aoclsparse_mi355_mm_state_adopt on each of the 7 other ranks byte for byte (header scalars and every buffer)."""
    import ctypes
    import numpy as np
    import torch

    pkg = entry.load_package()
    import aocl_sparse_amd.sharded as sharded

    def reference_split(n, t, T, bblk=4):  # csrmm_kt.cpp:71-76, restated here (not through the library's or bench.py's rule)
        start = n * t // T
        start = start if start % bblk == 0 else start + bblk - start % bblk
        start = min(start, n)
        end = n * (t + 1) // T
        end = end if end % bblk == 0 else end + bblk - end % bblk
        return start, min(end, n)

    class FakeLibA:
        def aoclsparse_set_mm_hint(self, *a):
            return 0

        def aoclsparse_optimize(self, *a):
            return 0

    class FakeMatrixA:
        status, m, n, nnz, h = 0, 1000, 1000, 4996, 1

        def __init__(self, *a):
            pass

    calls = []

    class FakePkgA:
        STATUS, OP_NONE, ORDER_COLUMN, ORDER_ROW = pkg.STATUS, pkg.OP_NONE, pkg.ORDER_COLUMN, pkg.ORDER_ROW
        Matrix = FakeMatrixA
        column_shard = staticmethod(pkg.column_shard)

        class Descr:
            h = 2

        def lib(self):
            return FakeLibA()

        def dcsrmm(self, op, alpha, A, descr, order, B, n, ldb, beta, C, ldc):
            calls.append((n, ldb, ldc, order))
            return 0

    real_bh = sharded.broadcast_handle
    sharded.broadcast_handle = lambda pkg_, torch_, dist_, device_, rank_, world_, A_: (FakeMatrixA(), 0.0, "mock")
    try:
        for ncols in (256, 250):
            cover = []
            for rank in range(8):
                sh = sharded.ShardedCsrmm(FakePkgA(), torch, None, "cpu", rank, 8, (1000, 1000, None, None, None), ncols, "row")
                assert (sh.j0, sh.j1) == reference_split(ncols, rank, 8), (ncols, rank, sh.j0, sh.j1)
                cover += list(range(sh.j0, sh.j1))
                calls.clear()
                assert sh.run(None, None) == 0
                assert calls == ([(sh.nloc, sh.nloc, sh.nloc, pkg.ORDER_ROW)] if sh.nloc else [])
            assert cover == list(range(ncols))
        assert [reference_split(250, t, 8) for t in range(8)] == [(0, 32), (32, 64), (64, 96), (96, 128), (128, 156), (156, 188),
                                                                  (188, 220), (220, 250)]
    finally:
        sharded.broadcast_handle = real_bh

    # (b) the state's bytes: rank 0 exports, ranks 1..7 adopt
    rng = np.random.default_rng(5)
    bufs = [np.frombuffer(rng.bytes(n), dtype=np.uint8).copy() if n else None
            for n in (4004, 19984, 39968, 0, 512, 0, 0, 96, 0, 0, 0, 0, 640, 72)][: pkg.MM_STATE_BUFFERS]
    scalars = [int(v) for v in rng.integers(0, 1 << 40, size=40)]
    adopted = {}

    class Wire:
        """rank 0's broadcasts in order; every other rank replays them"""
        sent = []

    class FakeDistB:
        class ReduceOp:
            MAX, SUM, MIN = "max", "sum", "min"

        def __init__(self, rank):
            self.rank, self.k = rank, 0

        def is_initialized(self):
            return True

        def get_backend(self):
            return "gloo"

        def get_world_size(self):
            return 8

        def barrier(self):
            pass

        def all_reduce(self, t, op=None):
            pass

        def broadcast(self, t, src):
            if self.rank == 0:
                Wire.sent.append(t.clone())
            else:
                t.copy_(Wire.sent[self.k])
            self.k += 1

    class FakeMatrixB:
        def __init__(self, rank):
            self.rank = rank

        def mm_state_export(self):
            st = pkg.MmState()
            for i in range(40):
                st.scalars[i] = scalars[i]
            for i in range(pkg.MM_STATE_BUFFERS):
                st.bytes[i] = len(bufs[i]) if i < len(bufs) and bufs[i] is not None else 0
            return 0, st, [b.ctypes.data if b is not None else None for b in bufs]

        @classmethod
        def mm_state_adopt(cls, state, ptrs, double=True):
            got = [bytes((ctypes.c_ubyte * state.bytes[i]).from_address(p)) if p else None for i, p in enumerate(ptrs)]
            adopted[cls.current] = (list(state.scalars), list(state.bytes), got)
            return 0, FakeMatrixB(cls.current)

    class FakePkgB:
        CommId, MM_STATE_BUFFERS, STATUS, MmState = pkg.CommId, pkg.MM_STATE_BUFFERS, pkg.STATUS, pkg.MmState
        Matrix = FakeMatrixB

        def lib(self):
            return None

    real_as_tensor = torch.as_tensor

    def as_tensor(obj, *a, **k):  # (the zero-copy device view of the library's buffer: here the bytes live in host memory)
        if isinstance(obj, sharded._DeviceView):
            ptr, nbytes = obj.__cuda_array_interface__["data"][0], obj.__cuda_array_interface__["shape"][0]
            return torch.frombuffer(bytearray((ctypes.c_ubyte * nbytes).from_address(ptr)), dtype=torch.uint8)
        return real_as_tensor(obj, *a, **k)

    torch.as_tensor = as_tensor
    sharded._LIB_COMM.update(tried=False, ok=False)
    try:
        for rank in range(8):
            FakeMatrixB.current = rank
            out, ms, how = sharded.broadcast_handle(FakePkgB(), torch, FakeDistB(rank), "cpu", rank, 8, FakeMatrixB(0) if rank == 0 else None)
            assert "torch.distributed (gloo)" in how and isinstance(out, FakeMatrixB)
    finally:
        torch.as_tensor = real_as_tensor
        sharded._LIB_COMM.update(tried=False, ok=False)
    assert sorted(adopted) == list(range(1, 8))
    want = [bytes(b) if b is not None else None for b in bufs]
    for rank in range(1, 8):
        sc, by, got = adopted[rank]
        assert sc == scalars and by[: len(bufs)] == [len(b) if b is not None else 0 for b in bufs]
        assert got == want, rank


def test_gather_scalars_identity_and_shape():
    import torch
    pkg = entry.load_package()
    import aocl_sparse_amd.sharded as sharded
    assert sharded.gather_scalars(1.5, None, torch, "cpu", 0, 1) == [1.5]
