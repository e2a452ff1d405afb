"""Column-sharded csrmm, one process per GPU (BASELINE.json configs[3]).

C = alpha * A * B + beta * C with a tall dense B: column j of C depends on A and on column j of B only, so B and C are
split into `world` column slabs by the reference's own rule (it splits B's columns over its worker threads,
library/src/level3/aoclsparse_csrmm_kt.cpp:68-82 -- here a rank takes the place of a thread;
`aoclsparse_mi355_column_shard`), every rank holds A, and the data path has NO collective.  Communication happens
twice, outside the product: A travels once from rank 0 -- in its DEVICE FORMAT, i.e. the CSR arrays in HBM plus every csrmm
plan rank 0's aoclsparse_optimize built, so that the other ranks do no analysis (`broadcast_handle`: over the library's own
RCCL communicator when the backend is "nccl", else over torch.distributed + aoclsparse_mi355_mm_state_adopt) -- and a caller
that wants all of C everywhere can all-gather the slabs afterwards (`gather_C`, reported separately).

The arithmetic is the library's (`aoclsparse_dcsrmm` through the C ABI); this module is host-side orchestration only and
holds no compute.  Works with backend "nccl" (one rank per GPU: the driver's 2/4/8-GPU runs) and with "gloo" (CPU
tensors on the wire: lets two processes share ONE GPU so the whole control flow is testable on a 1-GPU box).
"""
import os
import time

import numpy as np


def _backend(dist):
    return dist.get_backend() if dist is not None and dist.is_initialized() else None


def reduce_scalar(value, op, dist=None, device="cpu"):
    """max / sum / min of a python float over all ranks (identity when not distributed)."""
    if dist is None or not dist.is_initialized():
        return float(value)
    import torch

    dev = "cpu" if _backend(dist) == "gloo" else device
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op={"max": dist.ReduceOp.MAX, "sum": dist.ReduceOp.SUM, "min": dist.ReduceOp.MIN}[op])
    return float(t.item())


def communicator_info(dist, torch):
    """What the ranks actually talk through: backend, world size and -- for "nccl" -- the RCCL version torch was built
    against (torch.cuda.nccl.version() reports RCCL's on ROCm).  Goes into the bench line so that a multi-GPU record shows
    that RCCL saw N ranks."""
    if dist is None or not dist.is_initialized():
        return {"backend": None, "world": 1}
    info = {"backend": dist.get_backend(), "world": dist.get_world_size()}
    if info["backend"] == "nccl":
        try:
            info["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception as e:  # the version query is informative only
            info["rccl_version"] = "unavailable (%s)" % type(e).__name__
    return info


def gather_scalars(value, dist, torch, device, rank, world):
    """one python float per rank -> the list of all of them on every rank (an all-reduce of a one-hot vector)"""
    if dist is None or not dist.is_initialized() or world == 1:
        return [float(value)]
    wire = "cpu" if _backend(dist) == "gloo" else device
    t = torch.zeros(world, dtype=torch.float64, device=wire)
    t[rank] = float(value)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(v) for v in t.cpu().tolist()]


def broadcast_text(dist, torch, device, rank, text):
    """a short string from rank 0 to every rank (descriptions that only rank 0 can compose)"""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return text
    wire = "cpu" if _backend(dist) == "gloo" else device
    buf = torch.zeros(512, dtype=torch.uint8, device=wire)
    if rank == 0:
        raw = text.encode()[:511]
        buf[: len(raw)] = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(wire)
    dist.broadcast(buf, 0)
    return bytes(buf.cpu().numpy().tobytes()).split(b"\0", 1)[0].decode(errors="replace")


def barrier(dist, torch):
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    if dist is not None and dist.is_initialized():
        dist.barrier()
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def broadcast_csr(dist, torch, device, rank, csr):
    """csr = (m, n, row_ptr, col_ind, val) numpy arrays on rank 0 (anything on the others) -> the same five on every
    rank (host numpy arrays, which is what aoclsparse_create_dcsr aliases) + the broadcast time in ms.  With "nccl" the
    three arrays travel GPU to GPU (RCCL); with "gloo" as CPU tensors."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return csr, 0.0
    wire = "cpu" if _backend(dist) == "gloo" else device
    meta = torch.zeros(3, dtype=torch.int64, device=wire)
    if rank == 0:
        m, n, rp, ci, v = csr
        meta = torch.tensor([m, n, len(v)], dtype=torch.int64, device=wire)
    dist.broadcast(meta, 0)
    m, n, nnz = (int(t) for t in meta.tolist())
    if rank == 0:
        t_rp = torch.from_numpy(np.ascontiguousarray(rp, dtype=np.int32)).to(wire)
        t_ci = torch.from_numpy(np.ascontiguousarray(ci, dtype=np.int32)).to(wire)
        t_v = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float64)).to(wire)
    else:
        t_rp = torch.empty(m + 1, dtype=torch.int32, device=wire)
        t_ci = torch.empty(nnz, dtype=torch.int32, device=wire)
        t_v = torch.empty(nnz, dtype=torch.float64, device=wire)
    barrier(dist, torch)
    t0 = time.perf_counter()
    for t in (t_rp, t_ci, t_v):
        dist.broadcast(t, 0)
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    if rank == 0:
        return (m, n, rp, ci, v), ms
    return (m, n, t_rp.cpu().numpy(), t_ci.cpu().numpy(), t_v.cpu().numpy()), ms


class _DeviceView:
    """zero-copy view of `nbytes` bytes of device memory at `ptr` for torch.as_tensor (CUDA array interface, version 2)"""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


_LIB_COMM = {"tried": False, "ok": False}


def library_communicator(pkg, torch, dist, device, rank, world):
    """The library's own RCCL communicator (aoclsparse_mi355_comm_*), set up once per process for an "nccl" job: rank 0 draws
    the unique id, torch.distributed carries its 128 bytes, every rank joins.  All ranks agree on the outcome (an all-reduce of
    the status), so a rank on which librccl could not be loaded sends the whole job down the torch.distributed path instead of
    leaving the others inside a collective.  -> True when the communicator is usable on EVERY rank."""
    if _LIB_COMM["tried"]:
        return _LIB_COMM["ok"]
    _LIB_COMM["tried"] = True
    if _backend(dist) != "nccl" or world < 2:
        return False
    L = pkg.lib()
    cid = pkg.CommId()
    # ncclCommInitRank blocks until EVERY rank has called it: first make sure every rank CAN (librccl loads, an id can be drawn --
    # not a collective), and only then enter the collective
    st = L.aoclsparse_mi355_comm_unique_id(cid)
    if reduce_scalar(1.0 if st == 0 else 0.0, "min", dist, device) < 0.5:
        return False
    wire = torch.zeros(128, dtype=torch.uint8, device=device)
    if rank == 0:
        wire.copy_(torch.frombuffer(bytearray(bytes(cid)), dtype=torch.uint8))
    dist.broadcast(wire, 0)  # rank 0's id is the job's
    import ctypes
    ctypes.memmove(ctypes.addressof(cid), wire.cpu().numpy().tobytes(), 128)
    # ncclCommInitRank has never met N > 1 GPUs on the builder's boxes: it runs in a helper thread (ctypes releases the GIL) with a
    # deadline, so a rank that does not come back leaves the job on the torch.distributed wire instead of hanging it (the thread
    # is a daemon; a communicator only some ranks joined is never used)
    import threading
    box = {}
    th = threading.Thread(target=lambda: box.setdefault("st", L.aoclsparse_mi355_comm_init(world, rank, cid)), daemon=True)
    th.start()
    th.join(float(os.environ.get("AOCLSPARSE_MI355_COMM_INIT_TIMEOUT", "120")))
    ok = (not th.is_alive()) and box.get("st", 1) == 0
    ok = reduce_scalar(1.0 if ok else 0.0, "min", dist, device) > 0.5
    _LIB_COMM["ok"] = ok
    return ok


def broadcast_handle(pkg, torch, dist, device, rank, world, A):
    """A: rank 0's handle after set_mm_hint + aoclsparse_optimize (None elsewhere) -> (handle on every rank, ms, how).
    What travels is the analysed device state (CSR arrays in HBM + every csrmm plan); the receivers adopt it and run no
    analysis.  Wire: the library's RCCL communicator ("nccl" jobs), else torch.distributed (RCCL tensors with "nccl" when the
    library communicator is unavailable, CPU tensors with "gloo")."""
    if dist is None or not dist.is_initialized() or world == 1:
        return A, 0.0, "one rank: nothing to send"
    L = pkg.lib()
    barrier(dist, torch)
    t0 = time.perf_counter()
    if library_communicator(pkg, torch, dist, device, rank, world):
        import ctypes
        h = A.h if rank == 0 else ctypes.c_void_p()
        st = L.aoclsparse_mi355_comm_broadcast_matrix(ctypes.byref(h), 0)
        L.aoclsparse_mi355_synchronize()
        if reduce_scalar(1.0 if st == 0 else 0.0, "min", dist, device) > 0.5:
            out = A if rank == 0 else pkg.Matrix.from_handle(h)
            return out, (time.perf_counter() - t0) * 1e3, "library RCCL communicator: ncclBroadcast of the analysed device state"
        if rank != 0 and st == 0:
            L.aoclsparse_destroy(ctypes.byref(h))
    # torch.distributed as the wire
    cpu_wire = _backend(dist) == "gloo"
    wire = "cpu" if cpu_wire else device
    NB = pkg.MM_STATE_BUFFERS
    hdr = torch.zeros(40 + NB + 1, dtype=torch.int64, device=wire)
    ptrs = [None] * pkg.MM_STATE_BUFFERS
    if rank == 0:
        st, state, ptrs = A.mm_state_export()
        hdr[:40] = torch.tensor(list(state.scalars), dtype=torch.int64)
        hdr[40:40 + NB] = torch.tensor(list(state.bytes), dtype=torch.int64)
        hdr[40 + NB] = st
    dist.broadcast(hdr, 0)
    h = [int(x) for x in hdr.cpu().tolist()]
    if h[40 + NB] != 0:
        raise RuntimeError("aoclsparse_mi355_mm_state_export failed on rank 0: %s" % pkg.STATUS.get(h[40 + NB], h[40 + NB]))
    held = []
    for i in range(pkg.MM_STATE_BUFFERS):
        nbytes = h[40 + i]
        if nbytes == 0:
            held.append(None)
            continue
        if rank == 0:
            t = torch.as_tensor(_DeviceView(ptrs[i], nbytes), device=device)  # the library's buffer itself, no copy
            t = t.cpu() if cpu_wire else t
        else:
            t = torch.empty(nbytes, dtype=torch.uint8, device=wire)
        dist.broadcast(t, 0)
        held.append(t if rank == 0 else t.to(device))
    if rank == 0:
        out = A
    else:
        state = pkg.MmState()
        for i in range(40):
            state.scalars[i] = h[i]
        for i in range(pkg.MM_STATE_BUFFERS):
            state.bytes[i] = h[40 + i]
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        st, out = pkg.Matrix.mm_state_adopt(state, [t.data_ptr() if t is not None else None for t in held])
        assert st == 0, pkg.STATUS.get(st, st)
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    return out, (time.perf_counter() - t0) * 1e3, ("torch.distributed (%s): the analysed device state, adopted with "
                                                  "aoclsparse_mi355_mm_state_adopt" % _backend(dist))


def make_B_slab(torch, device, m, j0, j1, layout, seed=777):
    """Columns [j0, j1) of the job's B: column j is U(-1, 1) from a generator seeded seed + j, so any sharding of the
    same job sees the same matrix.  layout "col": slab stored column-major with ld = m; "row": m x (j1-j0), ld = j1-j0."""
    nloc = j1 - j0
    cols = torch.empty((nloc, m), dtype=torch.float64, device=device)
    gen = torch.Generator(device=device)
    for jl in range(nloc):
        gen.manual_seed(seed + j0 + jl)
        cols[jl].copy_(torch.rand(m, dtype=torch.float64, device=device, generator=gen) * 2.0 - 1.0)
    if layout == "col":
        return cols.reshape(-1)
    return cols.t().contiguous().reshape(-1)


class ShardedCsrmm:
    """One rank's share of C = alpha*A*B + beta*C.  Every rank builds the same handle from the broadcast CSR arrays
    (mm hint + aoclsparse_optimize), owns columns [j0, j1) and calls the ordinary aoclsparse_dcsrmm on its slab."""

    def __init__(self, pkg, torch, dist, device, rank, world, csr, ncols, layout="col"):
        assert layout in ("col", "row")
        self.pkg, self.torch, self.dist, self.device = pkg, torch, dist, device
        self.rank, self.world, self.ncols, self.layout = rank, world, ncols, layout
        L = pkg.lib()
        self.descr = pkg.Descr()
        A, self.optimize_ms = None, 0.0
        if rank == 0:
            # rank 0 alone creates the handle and analyses it (mm hint + aoclsparse_optimize) ...
            m, n, rp, ci, v = csr
            A = pkg.Matrix(0, m, n, rp, ci, v)
            assert A.status == 0, pkg.STATUS[A.status]
            st = L.aoclsparse_set_mm_hint(A.h, pkg.OP_NONE, self.descr.h, 100)
            assert st == 0, pkg.STATUS[st]
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            t0 = time.perf_counter()
            st = L.aoclsparse_optimize(A.h)
            assert st == 0, pkg.STATUS[st]
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            self.optimize_ms = (time.perf_counter() - t0) * 1e3
        # ... and its device format travels: the other ranks adopt it and analyse nothing
        self.A, self.a_broadcast_ms, self.a_broadcast_how = broadcast_handle(pkg, torch, dist, device, rank, world, A)
        self.m, self.n, self.nnz = self.A.m, self.A.n, self.A.nnz
        self.j0, self.j1 = pkg.column_shard(ncols, world, rank)
        self.nloc = self.j1 - self.j0

    def slab_ld(self, rows, nloc=None):
        nloc = self.nloc if nloc is None else nloc
        return rows if self.layout == "col" else max(nloc, 1)

    def make_B(self, seed=777, j0=None, j1=None):
        j0 = self.j0 if j0 is None else j0
        j1 = self.j1 if j1 is None else j1
        return make_B_slab(self.torch, self.device, self.n, j0, j1, self.layout, seed)

    def run(self, B, C, alpha=1.0, beta=0.0, nloc=None):
        """B, C: this rank's slabs (device tensors, layout as constructed)."""
        pkg = self.pkg
        nloc = self.nloc if nloc is None else nloc
        if nloc == 0:
            return 0
        order = pkg.ORDER_COLUMN if self.layout == "col" else pkg.ORDER_ROW
        return pkg.dcsrmm(pkg.OP_NONE, alpha, self.A, self.descr, order, B, nloc, self.slab_ld(self.n, nloc), beta, C,
                          self.slab_ld(self.m, nloc))

    def gather_C(self, C):
        """All slabs on every rank -> (full C as an (ncols, m) tensor of columns, ms).  Optional: the product itself
        never needs it."""
        C = C[: self.nloc * self.m]  # a rank that owns no columns (more ranks than 4-column blocks) holds a dummy buffer
        cols = C.reshape(self.nloc, self.m) if self.layout == "col" else C.reshape(self.m, self.nloc).t().contiguous()
        shards = [self.pkg.column_shard(self.ncols, self.world, r) for r in range(self.world)]
        return gather_slabs(self.torch, self.dist, self.device, self.rank, shards, cols, pkg=self.pkg)


def gather_slabs(torch, dist, device, rank, shards, cols, pkg=None):
    """cols: this rank's slab as a (width, m) tensor; shards: [(j0, j1)] of every rank -> ((ncols, m) tensor, ms).
    Equal slabs of an "nccl" job whose library communicator is up (library_communicator) go through
    aoclsparse_mi355_comm_allgather (ncclAllGather on the library's stream: the slabs land in place, in rank order);
    everything else through torch.distributed."""
    if dist is None or not dist.is_initialized() or len(shards) == 1:
        return cols, 0.0
    m = cols.shape[1]
    if pkg is not None and _LIB_COMM["ok"] and len({b - a for a, b in shards}) == 1 and cols.is_cuda:
        send = cols.contiguous()
        full = torch.empty((len(shards) * send.shape[0], m), dtype=cols.dtype, device=device)
        barrier(dist, torch)
        t0 = time.perf_counter()
        st = pkg.lib().aoclsparse_mi355_comm_allgather(send.data_ptr(), full.data_ptr(), send.numel() * send.element_size())
        pkg.lib().aoclsparse_mi355_synchronize()
        ms = (time.perf_counter() - t0) * 1e3
        if reduce_scalar(1.0 if st == 0 else 0.0, "min", dist, device) > 0.5:
            return full, ms
        # (a failed collective on some rank: fall through to torch.distributed on every rank)
    wire = "cpu" if _backend(dist) == "gloo" else device
    parts = [torch.empty((b - a, m), dtype=cols.dtype, device=wire) for a, b in shards]
    barrier(dist, torch)
    t0 = time.perf_counter()
    if len({b - a for a, b in shards}) == 1:
        dist.all_gather(parts, cols.to(wire).contiguous())
    else:  # slabs differ in width (n not a multiple of 4*world): broadcast each from its owner
        for r, (a, b) in enumerate(shards):
            if r == rank:
                parts[r].copy_(cols)
            if b > a:
                dist.broadcast(parts[r], r)
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    return torch.cat([p.to(device) for p in parts], dim=0), ms


def quartiles(ms):
    """min / q1 / median / q3 / max of per-iteration times -- the statistics the reference harness prints
    (tests/include/aoclsparse_stats.hpp:41-129)."""
    a = np.sort(np.asarray(ms, dtype=np.float64))
    if len(a) == 0:
        return None
    q = lambda f: float(np.quantile(a, f))
    return {"min": float(a[0]), "q1": q(0.25), "median": q(0.5), "q3": q(0.75), "max": float(a[-1]), "n": int(len(a))}


def csrmm_bytes(m, k, nnz, ncols, beta_nonzero=False):
    """Dense-correct csrmm byte count."""
    return (m + 1 + nnz) * 4 + nnz * 8 + 8 * ncols * (k + m * (2 if beta_nonzero else 1))


def bench_sharded_csrmm(pkg, torch, dist, device, rank, world, csr, ncols, layout="col", reps=20, warm=3,
                        full_product=True, allgather=True, peak_gbs=8000.0, c_is_read=True):
    """The BASELINE config-4 job on `world` ranks.  Returns (on every rank) a dict with the sharded time (max over
    ranks of the median per-iteration device time, and the barrier-bracketed wall clock), the 1-GPU time T1 of the SAME
    job measured in the same run (every rank runs all `ncols` columns once `full_product` is set), the scaling
    efficiency T1 / (world * Tg), the A broadcast and the optional C all-gather.  c_is_read: the library's default for
    beta = 0 reads C (the reference's arithmetic), so the byte model counts that read; pass False when the overwrite mode of
    aoclsparse_mi355_set_csrmm_beta0_overwrite is on."""
    sh = ShardedCsrmm(pkg, torch, dist, device, rank, world, csr, ncols, layout)
    m, nnz = sh.m, sh.nnz
    B = sh.make_B()
    C = torch.zeros(max(sh.nloc, 1) * m, dtype=torch.float64, device=device)

    def timed(fn, reps, warm):
        for _ in range(warm):
            assert fn() == 0
        barrier(dist, torch)
        t0 = time.perf_counter()
        pkg.timer_mark()
        for _ in range(reps):
            assert fn() == 0
            pkg.timer_mark()
        laps = pkg.timer_laps()
        barrier(dist, torch)
        return laps, (time.perf_counter() - t0) / reps * 1e3

    laps, wall_ms = timed(lambda: sh.run(B, C), reps, warm)
    st_shard = quartiles(laps)
    if sh.nloc == 0:  # nothing was launched here: this rank must not win the max with timer noise nor divide by ~0 below
        st_shard = {k: 0.0 for k in ("min", "q1", "median", "q3", "max")}
        st_shard["n"] = 0
    tg_dev = reduce_scalar(st_shard["median"], "max", dist, device)
    tg_wall = reduce_scalar(wall_ms, "max", dist, device)
    # every rank's own slab time (multi-GPU reporting: the spread shows a slow GPU or an uneven split)
    per_rank = gather_scalars(st_shard["median"], dist, torch, device, rank, world)
    checksum = reduce_scalar(float(C[: sh.nloc * m].sum().item()) if sh.nloc else 0.0, "sum", dist, device)
    out = {
        "layout": "column-major" if layout == "col" else "row-major", "ncols": ncols, "world": world,
        "cols_per_rank": sh.nloc, "m": m, "nnz": nnz, "c_is_read": bool(c_is_read),
        "shard_ms": st_shard, "slab_ms_per_rank": [round(v, 5) for v in per_rank],
        "tg_ms_device_median_max_over_ranks": round(tg_dev, 5),
        "tg_ms_wall_max_over_ranks": round(tg_wall, 5),
        "a_broadcast_ms": round(reduce_scalar(sh.a_broadcast_ms, "max", dist, device), 3), "a_broadcast_how": sh.a_broadcast_how,
        "setup_optimize_ms_rank0": round(reduce_scalar(sh.optimize_ms, "max", dist, device), 3),
        "checksum": checksum,
    }
    job_bytes = csrmm_bytes(m, sh.n, nnz, ncols, c_is_read) + (world - 1) * ((m + 1 + nnz) * 4 + nnz * 8)
    shard_bytes = csrmm_bytes(m, sh.n, nnz, sh.nloc, c_is_read)
    out["gflops_job"] = round(2.0 * nnz * ncols / tg_dev / 1e6, 2)
    shard_gbs = shard_bytes / st_shard["median"] / 1e6 if sh.nloc and st_shard["median"] > 0 else 0.0
    out["roofline_shard"] = {"bound": "hbm", "achieved": round(shard_gbs, 2), "peak": peak_gbs, "unit": "GB/s",
                             "frac": round(shard_gbs / peak_gbs, 4), "traffic": None,
                             "algorithmic_bytes_per_launch": shard_bytes if sh.nloc else 0}
    out["gbs_job_algorithmic"] = round(job_bytes / tg_dev / 1e6, 2)
    if allgather:
        full, ag_ms = sh.gather_C(C)
        out["c_allgather_ms"] = round(reduce_scalar(ag_ms, "max", dist, device), 3)
        out["c_allgather_checksum"] = float(full.sum().item())
        del full
    if full_product:
        if world == 1:
            t1, st_full = tg_dev, st_shard
        else:
            Bf = sh.make_B(j0=0, j1=ncols)
            Cf = torch.zeros(ncols * m, dtype=torch.float64, device=device)
            laps1, _ = timed(lambda: sh.run(Bf, Cf, nloc=ncols), max(3, reps // 2), 2)
            st_full = quartiles(laps1)
            t1 = reduce_scalar(st_full["median"], "min", dist, device)  # the fastest GPU's 1-GPU time: strictest T1
            del Bf, Cf
        out["t1_ms"] = round(t1, 5)
        out["full_ms"] = st_full
        out["efficiency"] = round(t1 / (world * tg_dev), 4)
        fb = csrmm_bytes(m, sh.n, nnz, ncols, c_is_read)
        out["roofline_full"] = {"bound": "hbm", "achieved": round(fb / t1 / 1e6, 2), "peak": peak_gbs, "unit": "GB/s",
                                "frac": round(fb / t1 / 1e6 / peak_gbs, 4), "traffic": None,
                                "algorithmic_bytes_per_launch": fb}
    return out, sh, B, C


# ---- row-sharded SpMV for iterative use ("next") ---------------------------------------------------------
# A single product does not shard (the all-gather of x costs more than the kernel), but an ITERATION that
# feeds y back as the next x (power iteration, Jacobi, the p of CG) can keep A split by rows: rank r owns rows [r0, r1) of A
# (all columns) and the matching slice of y; every product needs the FULL x, so each iteration ends with ONE all-gather of the
# slices (8 n bytes per rank pair over xGMI with "nccl").  Row-wise arithmetic does not depend on the split: every y_i has the
# bits of the unsharded product.


def row_shard(m, world, rank):
    """rows [r0, r1) of rank `rank`: contiguous, balanced to within one row"""
    return (m * rank) // world, (m * (rank + 1)) // world


def slice_rows(csr, r0, r1):
    """(m, n, row_ptr, col_ind, val) -> the same five for rows [r0, r1) (row pointers rebased to the slice's first entry;
    index base kept: row_ptr[0] of a base-b matrix stays b)"""
    m, n, rp, ci, v = csr
    base = int(rp[0])
    s, e = int(rp[r0]) - base, int(rp[r1]) - base
    return r1 - r0, n, (np.asarray(rp[r0:r1 + 1]) - s).astype(np.int32), np.ascontiguousarray(ci[s:e]), np.ascontiguousarray(v[s:e])


def allgather_rows(torch, dist, device, rank, world, m, y_loc, x_full):
    """x_full[0:m] <- concatenation of every rank's y_loc (lengths by row_shard); returns the time in ms.  The collective wants
    equal pieces: slices are padded to the longest one in a scratch tensor and compacted after the gather (m % world != 0)."""
    if dist is None or not dist.is_initialized() or world == 1:
        x_full[:m].copy_(y_loc[:m])
        return 0.0
    wire = "cpu" if _backend(dist) == "gloo" else device
    mx = max(row_shard(m, world, r)[1] - row_shard(m, world, r)[0] for r in range(world))
    r0, r1 = row_shard(m, world, rank)
    piece = torch.zeros(mx, dtype=y_loc.dtype, device=wire)
    piece[: r1 - r0].copy_(y_loc[: r1 - r0])
    pieces = torch.empty(world * mx, dtype=y_loc.dtype, device=wire)
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    dist.all_gather_into_tensor(pieces, piece)
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    for r in range(world):
        a, b = row_shard(m, world, r)
        x_full[a:b].copy_(pieces[r * mx: r * mx + (b - a)])
    return ms


class ShardedSpmv:
    """One rank's rows of y = A x (square A): handle over its row slice (mv hint + optimize), the full x on the device."""

    def __init__(self, pkg, torch, dist, device, rank, world, csr=None, build_rows=None, m=None):
        """csr: the whole matrix on rank 0 (broadcast, then sliced) -- or build_rows(r0, r1) -> (ml, n, row_ptr, col_ind, val):
        every rank builds ITS rows of the m x m matrix itself and no rank ever holds all of A (the form that scales)."""
        self.pkg, self.torch, self.dist, self.device, self.rank, self.world = pkg, torch, dist, device, rank, world
        if build_rows is not None:
            self.m = self.n = int(m)
            self.a_broadcast_ms = 0.0
            self.r0, self.r1 = row_shard(self.m, world, rank)
            ml, nl, rpl, cil, vl = build_rows(self.r0, self.r1)
            assert ml == self.r1 - self.r0 and nl == self.n
            self.nnz_loc = int(len(vl))
            self.nnz = int(reduce_scalar(self.nnz_loc, "sum", dist, device))
            base = int(rpl[0])
        else:
            (self.m, self.n, rp, ci, v), self.a_broadcast_ms = broadcast_csr(dist, torch, device, rank, csr)
            assert self.m == self.n, "the iteration feeds y back as x"
            self.r0, self.r1 = row_shard(self.m, world, rank)
            ml, nl, rpl, cil, vl = slice_rows((self.m, self.n, rp, ci, v), self.r0, self.r1)
            self.nnz_loc, self.nnz = int(len(vl)), int(len(v))
            base = int(rp[0])
        self.local = (ml, nl, rpl, cil, vl)
        self.A = pkg.Matrix(base, ml, nl, rpl, cil, vl)
        assert self.A.status == 0, pkg.STATUS[self.A.status]
        self.descr = pkg.Descr(base=base)
        L = pkg.lib()
        assert L.aoclsparse_set_mv_hint(self.A.h, pkg.OP_NONE, self.descr.h, 1000) == 0
        assert L.aoclsparse_optimize(self.A.h) == 0

    def product(self, x_full, y_loc, alpha=1.0, beta=0.0):
        return self.pkg.dmv(self.pkg.OP_NONE, alpha, self.A, self.descr, x_full, beta, y_loc)

    def gather(self, y_loc, x_full):
        return allgather_rows(self.torch, self.dist, self.device, self.rank, self.world, self.m, y_loc, x_full)


def bench_sharded_spmv(pkg, torch, dist, device, rank, world, csr, iters=20, warm=3, peak_gbs=8000.0, build_rows=None, m=None):
    """`iters` iterations x <- A x (each: the local product, then the all-gather of the slices) on `world` ranks; returns the
    per-iteration times (product: max over ranks of the median device time; gather: max over ranks of the median wall time),
    the whole-job GFLOP/s including the gathers, and the bits of rank 0's slice for the caller's parity check."""
    sh = ShardedSpmv(pkg, torch, dist, device, rank, world, csr, build_rows=build_rows, m=m)
    m = sh.m
    x = torch.from_numpy(np.sin(0.01 * np.arange(m))).to(device)
    x0 = x.clone()
    y = torch.zeros(max(sh.r1 - sh.r0, 1), dtype=torch.float64, device=device)
    # parity input: ONE product of the un-iterated x (iterating a Laplacian overflows nothing in 20 steps, but the check
    # wants a known x)
    assert sh.product(x0, y) == 0
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    y_first = y[: sh.r1 - sh.r0].clone()
    prod_ms, gat_ms = [], []
    barrier(dist, torch)
    t_all = time.perf_counter()
    for it in range(warm + iters):
        if it == warm:
            barrier(dist, torch)
            t_all = time.perf_counter()
        pkg.timer_start()
        assert sh.product(x, y) == 0
        p = pkg.timer_stop()
        g = sh.gather(y, x)
        if it >= warm:
            prod_ms.append(p), gat_ms.append(g)
        # keep the iterate bounded (a scalar scale on every rank's copy: not part of the timed pieces' meaning)
        x.mul_(0.125)
    barrier(dist, torch)
    wall = (time.perf_counter() - t_all) / iters * 1e3
    tp = reduce_scalar(float(np.median(prod_ms)), "max", dist, device)
    tg = reduce_scalar(float(np.median(gat_ms)), "max", dist, device)
    tw = reduce_scalar(wall, "max", dist, device)
    loc_bytes = (sh.r1 - sh.r0 + 1 + sh.nnz_loc) * 4 + (sh.r1 - sh.r0 + m + sh.nnz_loc) * 8
    gbs = loc_bytes / tp / 1e6 if tp > 0 else 0.0
    # a share that fits the 256 MiB Infinity Cache can run above the HBM roofline: such a figure is no evidence, so the HBM
    # roofline object is only reported for shares of at least 512 MB
    hbm_sized = loc_bytes >= (512 << 20)
    return {"what": "x <- A x iterated, A split by rows over %d rank(s), one all-gather of the slices per iteration" % world,
            "world": world, "m": m, "nnz": sh.nnz, "rows_per_rank": sh.r1 - sh.r0, "a_broadcast_ms": round(sh.a_broadcast_ms, 3),
            "product_ms_median_max_over_ranks": round(tp, 5), "allgather_ms_median_max_over_ranks": round(tg, 5),
            "iteration_ms_wall_max_over_ranks": round(tw, 5), "gflops_job_products_only": round(2.0 * sh.nnz / tp / 1e6, 2) if tp > 0 else 0.0,
            "gflops_job_with_gather": round(2.0 * sh.nnz / (tp + tg) / 1e6, 2) if tp + tg > 0 else 0.0,
            "allgather_bytes_per_rank": 8 * m,
            "note": ("a rank's share of the matrix (%d MB) fits the 256 MiB Infinity Cache: the product can run above the HBM roofline"
                     % (loc_bytes >> 20)) if loc_bytes < (256 << 20) * 1.5 else None,
            "shard_bytes": loc_bytes, "shard_gbs_algorithmic": round(gbs, 2),
            "roofline_shard": {"bound": "hbm", "achieved": round(gbs, 2), "peak": peak_gbs, "unit": "GB/s",
                               "frac": round(gbs / peak_gbs, 4), "traffic": None,
                               "algorithmic_bytes_per_launch": loc_bytes} if hbm_sized else None}, sh, y_first, x0


class ShardedSp2m:
    """C = A * B with A split by rows over the ranks and B on every rank.  A row of C depends on
    its row of A and on B only, so the slices are independent products (aoclsparse_sp2m on a handle over the rank's rows of A):
    no exchange on the data path.  What the ranks do share is the size of their slices -- one all-gather of `world` integers --
    so that each knows where its rows start in the global row_ptr of C."""

    def __init__(self, pkg, torch, dist, device, rank, world, m, build_rows, b_csr):
        """build_rows(r0, r1) -> (ml, k, row_ptr, col_ind, val): the rank's rows of the m x k matrix A (no rank holds all of A);
        b_csr = (k, n, row_ptr, col_ind, val): the whole of B, held by every rank."""
        self.pkg, self.torch, self.dist, self.device, self.rank, self.world = pkg, torch, dist, device, rank, world
        self.m = int(m)
        self.r0, self.r1 = row_shard(self.m, world, rank)
        ml, k, rpl, cil, vl = build_rows(self.r0, self.r1)
        kb, self.n, rpb, cib, vb = b_csr
        assert ml == self.r1 - self.r0 and k == kb
        self.local = (ml, k, rpl, cil, vl)
        self.A = pkg.Matrix(int(rpl[0]), ml, k, rpl, cil, vl)
        self.B = pkg.Matrix(int(rpb[0]), kb, self.n, rpb, cib, vb)
        assert self.A.status == 0 and self.B.status == 0
        self.dA, self.dB = pkg.Descr(base=int(rpl[0])), pkg.Descr(base=int(rpb[0]))
        self.nnz_a_loc = int(len(vl))

    def product(self):
        """-> (status, handle of this rank's rows of C); the caller destroys the handle"""
        import ctypes

        C = ctypes.c_void_p()
        st = self.pkg.lib().aoclsparse_sp2m(self.pkg.OP_NONE, self.dA.h, self.A.h, self.pkg.OP_NONE, self.dB.h, self.B.h,
                                            self.pkg.STAGE_FULL, ctypes.byref(C))
        return st, C

    def slice_offsets(self, nnz_loc):
        """all-gather of the slices' sizes -> (first entry of this rank's rows in the global C, total entries of C)"""
        if self.world == 1:
            return 0, int(nnz_loc)
        t = self.torch.tensor([int(nnz_loc)], dtype=self.torch.int64, device=self.device if _backend(self.dist) == "nccl" else "cpu")
        out = [self.torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        sizes = [int(o.item()) for o in out]
        return sum(sizes[: self.rank]), sum(sizes)


def bench_sharded_sp2m(pkg, torch, dist, device, rank, world, m, build_rows, b_csr, reps=5, warm=1):
    """`reps` products C_r = A_r * B on every rank (wall time of the call, host arrays out, as the single-GPU leg); returns the
    max over ranks of the median, the job's entries of C per second, and rank 0's slice for the caller's parity check."""
    import ctypes

    sh = ShardedSp2m(pkg, torch, dist, device, rank, world, m, build_rows, b_csr)
    L = pkg.lib()
    ts, keep = [], None
    barrier(dist, torch)
    for it in range(warm + reps):
        t = time.perf_counter()
        st, C = sh.product()
        dt = (time.perf_counter() - t) * 1e3
        assert st == 0, pkg.STATUS.get(st, st)
        if it >= warm:
            ts.append(dt)
        if it + 1 < warm + reps:
            L.aoclsparse_destroy(ctypes.byref(C))
        else:
            keep = C
    e = pkg.Matrix.from_handle(keep).export()
    nnz_loc = int(e["nnz"])
    first, total = sh.slice_offsets(nnz_loc)
    tn = reduce_scalar(float(np.median(ts)), "max", dist, device)
    barrier(dist, torch)
    res = {"what": "C = A * B, A split by rows over %d rank(s), B on every rank; independent slices, one all-gather of %d sizes"
                   % (world, world),
           "world": world, "m": sh.m, "n": sh.n, "rows_per_rank": sh.r1 - sh.r0, "nnz_c": total, "first_entry_of_this_rank": first,
           "product_ms_median_max_over_ranks": round(tn, 3), "entries_of_c_per_second_job": round(total / tn * 1e3, 1) if tn > 0 else 0.0}
    return res, sh, e
