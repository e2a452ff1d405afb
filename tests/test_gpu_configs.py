"""GPU tier (-m gpu), round 2: every BASELINE.json config at its stated size through the C ABI, the multi-rank csrmm
control flow on ONE GPU (two fresh processes, gloo on the wire), the csrmm kernels added this round, per-iteration
timing and the TRSV robustness cases.  Parity bars: bit-exact against the oracle wherever the reference's order is
reproduced; the componentwise bound of SURVEY.md section 8d (constant written in the test) where the schedule
legitimately differs (rows longer than one LDS tile in auto mode)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import oracle
from util import EPS64, ROOT, beta0_overwrite, laplace5, pkg, random_csr, trsv_schedule

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
P = pkg()
L = P.lib()
sys.path.insert(0, os.path.join(ROOT, "tools"))
import standins  # noqa: E402
import bench  # noqa: E402


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    assert torch.cuda.is_available(), "GPU tests need a GPU (no CPU fallback exists)"
    st, d, cus, name = P.device_info()
    assert st == 0 and cus > 0
    yield
    L.aoclsparse_mi355_set_pointer_mode(P.PTR_AUTO)


# --------------------------------------------------------------------------------------------------
# BASELINE configs[2]: the SuiteSparse mix at full size, kernel chosen by aoclsparse_optimize
# --------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["circuit-like", "web-like", "shell-like", "flan-like"])
def test_mix_full_size_dmv_after_optimize(name):
    """aoclsparse_set_mv_hint + aoclsparse_optimize + aoclsparse_dmv on the config-3 matrices (the real .mtx files when
    $MATRIX_DIR holds them, else the seeded stand-ins) against oracle.dcsrmv with the reference's dispatch rule
    (nnz <= 10 m -> scalar order, else the AVX-512 8-lane order, csrmv.hpp:326-343).
      * SELL-64 (chosen for the two uniform matrices) reproduces the dispatched order for rows of any length: bit-exact.
      * CSR-Adaptive (the two power-law matrices): rows of fewer than info.tree_min (32) entries bit-exact; a longer row is
        reduced by a wavefront tree in auto mode (round 5: inside an LDS tile too -- as one lane's chain a 300-entry row was the
        kernel's tail): |d| <= (2 ceil(log2 n) + 4 + n/256) eps sum|a x|.  aoclsparse_mi355_set_option(spmv_strict, 1) keeps
        every row in the reference's order: bit-exact everywhere."""
    label, m, rp, ci, v = standins.load(name)
    nnz = len(v)
    A = P.Matrix(0, m, m, rp, ci, v)
    d = P.Descr()
    assert L.aoclsparse_set_mv_hint(A.h, P.OP_NONE, d.h, 100) == 0 and L.aoclsparse_optimize(A.h) == 0
    info = A.spmv_info()
    uniform = name in ("shell-like", "flan-like")
    # uniform rows -> SELL-64; the mesh stand-ins (5 / 3 dofs per node) additionally share their column lists (kernel 4)
    assert info.kernel == (4 if uniform else 1), "optimize chose kernel %d for %s" % (info.kernel, label)
    assert info.order == (2 if nnz > 10 * m else 0)  # the reference's dispatch rule
    x = np.random.default_rng(1).uniform(-1, 1, m)
    y0 = np.random.default_rng(2).uniform(-1, 1, m)
    for alpha, beta in ((1.0, 0.0), (-0.75, 1.5)):
        yd = dev(y0)
        assert P.dmv(P.OP_NONE, alpha, A, d, dev(x), beta, yd) == 0
        torch.cuda.synchronize()
        got = yd.cpu().numpy()
        so, yr = oracle.dcsrmv(-1, 0, alpha, m, nnz, v, ci, rp, x, beta, y0, nthreads=oracle.max_threads())
        assert so == 0
        lens = np.diff(rp)
        if info.kernel in (3, 4):
            assert np.array_equal(got, yr)
        else:
            assert info.tree_min == 32
            short = lens < info.tree_min
            assert np.array_equal(got[short], yr[short])
            scale = np.zeros(m)
            nz = lens > 0
            scale[nz] = np.add.reduceat(np.abs(v * x[ci]), rp[:-1][nz])
            bound = (2 * np.ceil(np.log2(np.maximum(lens, 2))) + 4 + lens / 256.0) * EPS64 * abs(alpha) * scale
            assert np.all(np.abs(got - yr)[~short] <= bound[~short] + 2 * EPS64 * np.abs(beta * y0[~short]))
            assert (lens > info.tile).sum() == info.long_rows
            # strict mode: the reference's order for every row, long ones included
            assert L.aoclsparse_mi355_set_option(P.OPTION_SPMV_STRICT, 1) == 0
            try:
                assert A.spmv_info().tree_min == 0
                ys = dev(y0)
                assert P.dmv(P.OP_NONE, alpha, A, d, dev(x), beta, ys) == 0
                torch.cuda.synchronize()
                assert np.array_equal(ys.cpu().numpy(), yr)
            finally:
                assert L.aoclsparse_mi355_set_option(P.OPTION_SPMV_STRICT, 0) == 0


# --------------------------------------------------------------------------------------------------
# BASELINE configs[4]: unit-lower ILU(0) factor of the shell-like matrix through aoclsparse_dtrsv
# --------------------------------------------------------------------------------------------------
def _shell_factor():
    label, m, rp, ci, v = standins.load("shell-like")
    st, lu, dg = oracle.dilu0(m, 0, rp, ci, v)
    assert st == 0
    o = oracle.dcsr_optimize(m, m, len(lu), 0, rp, ci, lu)
    assert o["status"] == 0
    return label, m, rp, ci, lu, o


def test_shell_like_ilu0_trsv_full_size():
    """config 5 as BASELINE.md section 3 states it: ILU(0) of the af_shell10-like matrix (1.5 M rows, 25.6 M strict-lower
    entries, 5,505 dependency levels), descr {triangular, lower, unit}, alpha = 1, b = L * 1.  Every schedule returns
    the bits of the serial reference chain (ref_trsv_l, trsv_kr.hpp:57-75); the solution is the vector of ones to a
    few ulps and the residual ||L x - b||_inf / ||b||_inf is at rounding level."""
    label, m, rp, ci, lu, o = _shell_factor()
    A = P.Matrix(0, m, m, rp, ci, lu)
    dl = P.Descr(mtype=P.TYPE_TRIANGULAR, fill=P.FILL_LOWER, diag=P.DIAG_UNIT)
    assert L.aoclsparse_set_sv_hint(A.h, P.OP_NONE, dl.h, 10) == 0 and L.aoclsparse_optimize(A.h) == 0
    lv = A.trsv_levels(P.FILL_LOWER)
    assert lv > 1000
    ones = np.ones(m)
    _, b = oracle.dcsrmv_special("tri", 0, 1.0, m, m, 1, 0, lu, ci, rp, o["idiag"], o["iurow"], ones, 0.0, np.zeros(m))
    st, xr = oracle.dtrsv("l", 1.0, m, 0, lu, ci, rp, o["idiag"], b, True)
    assert st == 0
    bd = dev(b)
    for sched in (-1, 0, 1, 2, 3, 4, 5):  # the automatic choice, then every schedule (the kid selects the ARITHMETIC: round 3)
        with trsv_schedule(P, sched):
            xd = torch.full((m,), np.nan, dtype=torch.float64, device="cuda")
            assert P.dtrsv(P.OP_NONE, 1.0, A, dl, bd, xd) == 0
            torch.cuda.synchronize()
            x = xd.cpu().numpy()
            assert np.array_equal(x, xr), "schedule %d differs from the serial chain" % sched
            if sched in (-1, 5):  # round 6: the two-level schedule is what the plan-time model picks for this factor
                info = A.trsv_info(P.FILL_LOWER)
                assert info.schedule == 5 and info.chunks > 50 and info.model_chunk_us < info.model_block_us, (info.schedule, info.chunks)
    # kid 3 = the arithmetic of kt_trsv_l with 512-bit vectors (what an AVX-512 host runs): bit-identical to its restatement
    st, xk = oracle.trsv_kt("l", 8, 1.0, m, 0, lu, ci, rp, o["idiag"], b, True)
    xd = torch.full((m,), np.nan, dtype=torch.float64, device="cuda")
    assert st == 0 and P.dtrsv(P.OP_NONE, 1.0, A, dl, bd, xd, kid=3) == 0
    torch.cuda.synchronize()
    assert np.array_equal(xd.cpu().numpy(), xk) and not np.array_equal(xk, xr)
    assert np.max(np.abs(x - 1.0)) <= 64 * EPS64
    _, lx = oracle.dcsrmv_special("tri", 0, 1.0, m, m, 1, 0, lu, ci, rp, o["idiag"], o["iurow"], x, 0.0, np.zeros(m))
    assert np.max(np.abs(lx - b)) <= 16 * EPS64 * np.max(np.abs(b))


# --------------------------------------------------------------------------------------------------
# BASELINE configs[3], multi-rank: two fresh processes on the one GPU, gloo on the wire
# --------------------------------------------------------------------------------------------------
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _torchrun(nproc, script, *args, timeout=600):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, script)] + list(args)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, "rc=%d\nstdout:\n%s\nstderr:\n%s" % (r.returncode, r.stdout[-3000:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert lines, r.stdout[-2000:]
    return json.loads(lines[-1])


@pytest.mark.parametrize("layout,cols,beta", [("col", 40, 0.0), ("row", 64, 0.0), ("row", 42, -1.25), ("col", 10, 2.0)])
def test_sharded_csrmm_two_processes_one_gpu(layout, cols, beta):
    """The whole N > 1 control flow of aocl-sparse_amd/sharded.py (A broadcast from rank 0, per-rank B slab, the product,
    the optional all-gather) in two fresh processes that share the one GPU: each rank's slab and the gathered C must
    equal the single-rank product bit for bit, and so must the in-library aoclsparse_mi355_dcsrmm_shard entry."""
    res = _torchrun(2, "tools/sharded_check.py", "--backend", "gloo", "--grid", "150", "--cols", str(cols), "--layout", layout,
                    "--beta", str(beta))
    assert res["world"] == 2 and res["backend"] == "gloo"
    assert res["slab_bit_exact"] and res["gathered_bit_exact"] and res["abi_shard_bit_exact"] and res["all_ranks_ok"]
    assert res["shard"] == list(P.column_shard(cols, 2, 0))


def _bench_record(stdout, record_file):
    """bench.py's contract: the LAST stdout line is the short strict-JSON record (what the driver parses from a bounded tail of
    the output); the full report sits in the --record file.  Returns (short, full)."""
    lines = stdout.splitlines()
    assert lines and lines[-1].startswith("{"), stdout[-500:]
    assert len(lines[-1]) <= bench.COMPACT_LIMIT
    tail = stdout[-8192:]  # what a bounded tail of the output still holds
    short = json.loads(tail[tail.rstrip("\n").rfind("\n") + 1:])
    assert short == json.loads(lines[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "dtype", "config", "roofline", "cpu_baseline",
              "parity", "legs"):
        assert k in short, k
    for k in ("achieved", "peak", "frac", "traffic", "algorithmic_bytes_per_launch", "kernel_ms"):
        assert k in short["roofline"], k
    with open(record_file) as f:
        full = json.load(f)
    assert full["value"] == short["value"] and full["roofline"]["frac"] == short["roofline"]["frac"]
    return short, full


def test_bench_two_ranks_gloo_one_gpu(tmp_path):
    """bench.py as the driver starts it for N = 2 (torch.distributed.run, one process per rank), with --backend gloo so
    that both ranks can share the one GPU: the record must carry the whole-job value, the sharded csrmm objects (both layouts)
    with their efficiency T1 / (N TN), the A broadcast, the C all-gather and the parity verdicts."""
    rec = str(tmp_path / "legs.json")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "5",
           "--warmup", "2", "--grid", "512", "--mm-grid", "200", "--mm-cols", "64", "--shard-grid", "301", "--bell-nodes", "6", "--legs",
           "csrmm_sharded,spmv_row_sharded,sp2m_row_sharded", "--sp2m-grid", "150", "--record", rec]
    r = subprocess.run(cmd, cwd=ROOT, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4"), capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, "rc=%d\nstdout:\n%s\nstderr:\n%s" % (r.returncode, r.stdout[-3000:], r.stderr[-3000:])
    short, res = _bench_record(r.stdout, rec)
    _check_two_rank_record(short, res, own_rows=False)
    # the ranks' row slices of A * A are independent products; rank 0's slice against the oracle, the slices'
    # sizes all-gathered (the total is the unsharded product's: 5 * g^2 - 4 g entries of A give 13 g^2 - 36 g + 20 ... checked
    # against the oracle's count of the whole product)
    s2 = res["sp2m_row_sharded"]
    assert "error" not in s2, s2
    assert s2["world"] == 2 and s2["rows_per_rank"] == 150 * 150 // 2 and s2["parity"]["bit_exact"] is True
    import oracle
    from util import laplace5
    mg, rpg, cig, vg = laplace5(150)
    so, pcg, _, _ = oracle.dcsr2m(mg, mg, 0, rpg, cig, vg, 0, rpg, cig, vg)
    assert so == 0 and s2["nnz_c"] == int(pcg[mg]) and s2["first_entry_of_this_rank"] == 0
    assert short["legs"]["sp2m_row_sharded"]["parity"] is True


def test_bench_one_rank_under_the_launcher_with_the_nccl_backend(tmp_path):
    """bench.py under torch.distributed.run with its default backend ("nccl" = RCCL), one rank: what the driver's scaling run does
    at every N, here at the only N one GPU allows -- the process group is an RCCL communicator, the record says so (backend, world,
    RCCL version), the collective legs run (their exchanges are identities at world 1) and return the reference's bits."""
    rec = str(tmp_path / "legs.json")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2",
           "--grid", "512", "--mm-grid", "200", "--mm-cols", "64", "--shard-grid", "301", "--bell-nodes", "6", "--legs",
           "csrmm_sharded,spmv_row_sharded,sp2m_row_sharded", "--sp2m-grid", "150", "--record", rec]
    r = subprocess.run(cmd, cwd=ROOT, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4"), capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, "rc=%d\nstdout:\n%s\nstderr:\n%s" % (r.returncode, r.stdout[-3000:], r.stderr[-3000:])
    short, res = _bench_record(r.stdout, rec)
    comm = short["config"]["communicator"]
    assert comm["backend"] == "nccl" and comm["world"] == 1 and comm["rccl_version"][0].isdigit()
    assert res["n_gpus"] == 1 and res["parity"]["bit_exact"] is True
    for leg in ("csrmm_sharded_col", "csrmm_sharded_row", "csrmm_sharded_bell", "spmv_row_sharded", "sp2m_row_sharded"):
        assert "error" not in res[leg], (leg, res[leg])
        assert res[leg]["world"] == 1 and res[leg]["parity"]["bit_exact"] is True, (leg, res[leg])
        assert leg not in short["legs"]  # (round 6: the short record carries the sharded objects of runs with more than one rank only)


def _check_two_rank_record(short, res, own_rows):
    assert res["n_gpus"] == 2 and res["steps"] == 5 and res["scaling"] == "weak" and res["unit"] == "GFLOP/s"
    assert res["parity"]["bit_exact"] is True
    assert res["roofline"]["bound"] == "hbm" and 0 < res["roofline"]["frac"] < 1.0
    assert res["stats"]["n"] == 5 and res["stats"]["min"] <= res["stats"]["median"] <= res["stats"]["max"]
    assert short["cpu_baseline"] is None  # rank 0 at N = 1 only
    for lay, name in (("col", "column-major"), ("row", "row-major")):
        mm = res["csrmm_sharded_" + lay]
        assert "error" not in mm, mm
        assert mm["layout"] == name and mm["world"] == 2 and mm["cols_per_rank"] == 32 and mm["parity"]["bit_exact"] is True
        assert mm["efficiency"] > 0 and mm["t1_ms"] > 0 and mm["a_broadcast_ms"] > 0 and mm["c_allgather_ms"] > 0
        # beta = 0 reads C by default (the reference's arithmetic), so the byte model counts B, the read of C and its write
        assert mm["c_is_read"] is True
        assert mm["roofline_shard"]["algorithmic_bytes_per_launch"] == (40000 + 1 + mm["nnz"]) * 4 + mm["nnz"] * 8 + 8 * 32 * 3 * 40000
        assert short["legs"]["csrmm_sharded_" + lay]["efficiency"] == mm["efficiency"]
        assert short["legs"]["csrmm_sharded_" + lay]["parity"] is True
    # configs[3] to the letter: the block-dense stand-in on the blocked-ELL MFMA kernel, column slabs; the adopting rank runs the
    # MFMA kernel on the state it received (bell_width is read from every rank's handle)
    bl = res["csrmm_sharded_bell"]
    assert "error" not in bl, bl
    assert bl["layout"] == "column-major" and bl["world"] == 2 and bl["cols_per_rank"] == 32 and bl["bell_width"] == 7
    assert bl["parity"]["bit_exact"] is True and bl["efficiency"] > 0 and bl["m"] == 16 * 216
    assert short["legs"]["csrmm_sharded_bell"]["parity"] is True
    assert res["config"]["communicator"] == {"backend": "gloo", "world": 2} == short["config"]["communicator"]
    sp = res["spmv_row_sharded"]
    assert "error" not in sp, sp
    assert sp["world"] == 2 and sp["parity"]["bit_exact"] is True
    assert sp["product_ms_median_max_over_ranks"] > 0 and sp["allgather_ms_median_max_over_ranks"] > 0
    if not own_rows:
        # the row-sharded SpMV iteration (SURVEY 8e "next"): 301^2 = 90,601 rows over two ranks, an all-gather per iteration
        assert sp["m"] == 90601 and sp["rows_per_rank"] == 45300 and sp["allgather_bytes_per_rank"] == 8 * 90601
        assert sp["roofline_shard"] is None  # a cache-resident share carries no HBM roofline claim
    assert short["legs"]["spmv_row_sharded"]["parity"] is True


def test_bench_gpus_2_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2 ...` WITHOUT torch.distributed.run (the driver's command form): the parent starts the two ranks
    as a child process, relays rank 0's record as its own last line and exits with the child's code."""
    rec = str(tmp_path / "legs.json")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "5", "--warmup", "2",
                        "--grid", "512", "--mm-grid", "200", "--mm-cols", "64", "--shard-grid", "301", "--bell-nodes", "6", "--legs",
                        "csrmm_sharded,spmv_row_sharded", "--record", rec], cwd=ROOT, env=dict(env, OMP_NUM_THREADS="4"),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, "rc=%d\nstdout:\n%s\nstderr:\n%s" % (r.returncode, r.stdout[-3000:], r.stderr[-3000:])
    short, res = _bench_record(r.stdout, rec)
    _check_two_rank_record(short, res, own_rows=False)
    # the row-sharded SpMV leg as the multi-GPU run takes it: every rank builds ITS rows, no rank holds the whole matrix
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "5", "--warmup", "2",
                        "--grid", "512", "--shard-grid", "301", "--shard-own-rows", "--legs", "spmv_row_sharded", "--record", rec],
                       cwd=ROOT, env=dict(env, OMP_NUM_THREADS="4"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, "rc=%d\nstdout:\n%s\nstderr:\n%s" % (r.returncode, r.stdout[-3000:], r.stderr[-3000:])
    short, res = _bench_record(r.stdout, rec)
    sp = res["spmv_row_sharded"]
    assert "error" not in sp, sp
    assert sp["world"] == 2 and sp["m"] == 90601 and sp["rows_per_rank"] == 45300 and sp["nnz"] == 5 * 90601 - 4 * 301
    assert sp["parity"]["bit_exact"] is True and "each rank builds its own rows" in sp["workload"]
    # with RCCL two ranks cannot share the one GPU: the child fails and the parent must report that, not a record
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--grid", "256",
                        "--legs", "none", "--record", ""], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    import torch
    if torch.cuda.device_count() < 2:
        assert r.returncode != 0 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_bench_single_process_small_legs(tmp_path):
    """`python bench.py` with every leg on small inputs: the short record as the last line, and a full report whose legs all
    carry a roofline object and a bit-exact verdict (the driver-timed run uses the full sizes)."""
    rec = str(tmp_path / "legs.json")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "2", "--grid", "512",
                        "--mm-grid", "200", "--mm-cols", "64", "--shard-grid", "300", "--bell-nodes", "6", "--small", "--cpu-seconds", "0.5",
                        "--record", rec], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    short, res = _bench_record(r.stdout, rec)
    assert res["parity"]["bit_exact"] and res["cpu_baseline"]["kind"] == "port"
    cb = res["cpu_baseline"]
    assert cb["one_thread"]["threads"] == 1 and cb["cores"] >= 1 and cb["cpu_model"] and cb["bit_exact_vs_gpu"]
    assert short["cpu_baseline"]["kind"] == "port" and short["cpu_baseline"]["value"] == cb["value"] and short["cpu_baseline"]["cores"] == cb["cores"]
    assert res["l100"]["bit_exact"] and res["l100"]["roofline"]["frac"] > 0
    legs = res["legs"]
    for k in ("dcsrmv_csr_adaptive", "mix", "csrmm", "trsv"):
        assert k in legs and "error" not in legs[k], (k, legs.get(k))
    assert legs["dcsrmv_csr_adaptive"]["bit_exact_vs_headline_y"] and legs["dcsrmv_csr_adaptive"]["roofline"]["frac"] > 0
    for row in legs["mix"]["matrices"]:
        assert row["bit_exact_rows_below_tree_min"] and row["long_rows_within_bound"] and row["roofline"]["frac"] > 0
    kid_cases = 0
    for c in legs["csrmm"]["cases"]:
        assert c["roofline"]["frac"] > 0
        if c["mode"].startswith("kid"):  # pinned kid: checked against the KT restatement (8 columns) where the width allows
            kid_cases += 1
            assert c.get("bit_exact_8_columns_vs_kt_oracle", True), c["mode"]
        else:
            assert c["bit_exact_4_columns"], c["mode"]
    assert kid_cases >= 2
    for s in legs["trsv"]["schedules"]:
        assert s["bit_exact_vs_cpu"] and s["residual_inf"] < 1e-13
    for lay in ("col", "row", "bell"):
        assert res["csrmm_sharded_" + lay]["efficiency"] == 1.0 and res["csrmm_sharded_" + lay]["parity"]["bit_exact"]
    assert res["csrmm_sharded_bell"]["bell_width"] == 7
    assert res["spmv_row_sharded"]["parity"]["bit_exact"] and res["spmv_row_sharded"]["allgather_ms_median_max_over_ranks"] == 0.0
    # one number per leg in the short record, none of them missing and no leg in error
    n = short["legs"]
    assert "errors" not in n, n
    for k in ("l100_us", "csr_adaptive_frac", "mix_frac_mean", "mix_cold_frac_mean", "csrmm_row_ms", "csrmm_row_slab_ms", "csrmm_row_slab_frac",
              "csrmm_row_frac", "csrmm_row_eff8_cold", "trsv_us_per_block_level", "csrmm_col_ms", "csrmm_col_slab_ms", "csrmm_col_slab_frac", "csrmm_row_overwrite_slab_frac",
              "csrmm_col_overwrite_slab_frac", "trsv_ms"):
        assert n.get(k) is not None and n[k] > 0, (k, n)
    assert n["csrmm_parity"] and n["mix_parity"] and n["trsv_parity"]
    # round 6: the headline is the COLD product (cache flushed before each); the back-to-back loop keeps its own names; every mix
    # matrix says what bounds its back-to-back loop
    assert short["value_back_to_back"] > 0 and short["roofline"]["frac_back_to_back"] > 0
    assert abs(short["roofline"]["kernel_ms"] - short["ms_per_step"]) < 2e-6 and res["timing"]["wall_ms_per_step_including_the_flush"] > short["ms_per_step"]
    assert set(n["mix_bound"].values()) <= {"latency", "infinity_cache", "hbm"} and len(n["mix_cold_frac"]) == len(n["mix_frac"])
    # the twins of the headline product (round 5): fp32, the literal host-pointer call (PCIe inside), the same product with the
    # Infinity Cache flushed before every call, the launch-bound case as a C caller / inside a HIP graph sees it
    tw = legs["headline_twins"]
    assert "error" not in tw and tw["smv"]["bit_exact_first_2e20_rows"] and tw["host_pointer_dmv"]["bit_exact_vs_headline_y"]
    assert n["smv_parity"] and n["smv_frac"] > 0 and n["host_ptr_dmv_ms"] > short["ms_per_step"]
    assert n["dmv_cold_ms"] == tw["cold_dmv"]["ms"] > 0
    assert n["l100_graph_us"] > 0 and (n.get("l100_c_caller_us") is None or n["l100_c_caller_us"] > 0)


# --------------------------------------------------------------------------------------------------
# csrmm kernels of this round
# --------------------------------------------------------------------------------------------------
def _col_reference(alpha, base, v, ci, rp, m, k, Brm, n, ldb, beta, C0, ldc):
    """per-element reference bits for ROW-major operands: the column-major oracle on the transposed layouts"""
    Bc = np.ascontiguousarray(Brm.reshape(k, ldb)[:, :n].T).ravel()
    Cc = np.ascontiguousarray(C0.reshape(m, ldc)[:, :n].T).ravel()
    so, Cref = oracle.dcsrmm("col", alpha, base, v, ci, rp, m, Bc, n, k, beta, Cc, m)
    assert so == 0
    return Cref.reshape(n, m).T


@pytest.mark.parametrize("nx,ny", [(301, 75), (1000, 9)])
def test_csrmm_strip_order_banded_bit_exact(nx, ny):
    """Banded stencils (csrmm_api.cpp detect_row_runs: band >= 256, m >= 4 * band): the row-run kernel walks its 8-row
    blocks strip by strip (MmGroups::run_order).  A band that is not a multiple of the block, a matrix that is not a
    multiple of it either, the last strip narrower than the others: every row still comes out once, bit for bit."""
    m = nx * ny
    i = np.arange(m)
    cand = np.stack([i - nx, i - 1, i, i + 1, i + nx], axis=1)
    ok = (cand >= 0) & (cand < m)
    rp = np.concatenate([[0], np.cumsum(ok.sum(axis=1))]).astype(np.int32)
    ci = cand[ok].astype(np.int32)
    rng = np.random.default_rng(nx)
    v = rng.uniform(-1, 1, len(ci))
    A = P.Matrix(0, m, m, rp, ci, v)
    d = P.Descr()
    n = 128
    Br = rng.uniform(-1, 1, m * n)
    Cd = dev(np.full(m * n, 7.0))
    assert P.dcsrmm(P.OP_NONE, 1.0, A, d, P.ORDER_ROW, dev(Br), n, n, 0.0, Cd, n) == 0
    torch.cuda.synchronize()
    got = Cd.cpu().numpy().reshape(m, n)
    ref = _col_reference(1.0, 0, v, ci, rp, m, m, Br, n, n, 0.0, np.full(m * n, 7.0), n)
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("base", [0, 1])
def test_csrmm_row_runs_stencil_bit_exact(base):
    """csrmm_row_run_kernel (row-major, n >= 128, chosen when most rows repeat the previous row's column list shifted by
    one): a 7-point 3-D stencil with empty rows, a few rows of 9-40 entries (the plain loop), rows that break the pattern,
    a last partial run, padded leading dimensions, alpha / beta classes, NaN in C with beta = 0 (overwritten, as
    documented), float."""
    g = 22
    m = g * g * g
    rng = np.random.default_rng(31)
    rows = []
    for i in range(m):
        c = [i + o for o in (-g * g, -g, -1, 0, 1, g, g * g) if 0 <= i + o < m]
        if i % 311 == 7:
            c = []                                         # empty row
        elif i % 523 == 11:
            c = sorted(set(c) | set(int(t) for t in rng.integers(0, m, size=int(rng.integers(3, 34)))))  # long row
        elif i % 97 == 5:
            c = c[:-1]                                     # breaks the shifted pattern
        rows.append(np.array(c, dtype=np.int64))
    lens = np.array([len(c) for c in rows])
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32) + base
    ci = (np.concatenate(rows) + base).astype(np.int32)
    v = rng.uniform(-1, 1, len(ci))
    A = P.Matrix(base, m, m, rp, ci, v)
    d = P.Descr(base=base)
    for n, alpha, beta in ((128, 1.0, 0.0), (256, -0.5, 2.0), (130, 3.0, -1.0), (200, 1.0, 0.0)):
        ldb, ldc = n + 2, n + 4
        Br, C0 = rng.uniform(-1, 1, m * ldb), rng.uniform(-1, 1, m * ldc)
        Cd = dev(C0)
        assert P.dcsrmm(P.OP_NONE, alpha, A, d, P.ORDER_ROW, dev(Br), n, ldb, beta, Cd, ldc) == 0
        torch.cuda.synchronize()
        got = Cd.cpu().numpy().reshape(m, ldc)
        ref = _col_reference(alpha, base, v, ci, rp, m, m, Br, n, ldb, beta, C0, ldc)
        assert np.array_equal(got[:, :n], ref), "n=%d" % n
        assert np.array_equal(got[:, n:], C0.reshape(m, ldc)[:, n:])  # padding untouched
    n = 128
    Br = rng.uniform(-1, 1, m * n)
    Cd = torch.full((m * n,), float("nan"), dtype=torch.float64, device="cuda")
    with beta0_overwrite(P):  # opt-in: beta = 0 does not read C
        assert P.dcsrmm(P.OP_NONE, 1.0, A, d, P.ORDER_ROW, dev(Br), n, n, 0.0, Cd, n) == 0
        torch.cuda.synchronize()
    ref = _col_reference(1.0, base, v, ci, rp, m, m, Br, n, n, 0.0, np.zeros(m * n), n)
    got = Cd.cpu().numpy().reshape(m, n)
    nonempty = lens > 0
    assert np.array_equal(got[nonempty], ref[nonempty])
    # default: 0 * C is computed as in the reference (csrmm.hpp:83), so the NaN stays
    Cd = torch.full((m * n,), float("nan"), dtype=torch.float64, device="cuda")
    assert P.dcsrmm(P.OP_NONE, 1.0, A, d, P.ORDER_ROW, dev(Br), n, n, 0.0, Cd, n) == 0
    torch.cuda.synchronize()
    assert bool(torch.isnan(Cd).all())
    # float, same matrix
    vf = v.astype(np.float32)
    Af = P.Matrix(base, m, m, rp, ci, vf)
    Bf = rng.uniform(-1, 1, m * n).astype(np.float32)
    Cf = torch.zeros(m * n, dtype=torch.float32, device="cuda")
    assert L.aoclsparse_scsrmm(P.OP_NONE, 1.0, Af.h, d.h, P.ORDER_ROW, P._ptr(dev(Bf)), n, n, 0.0, P._ptr(Cf), n) == 0
    torch.cuda.synchronize()
    reff = np.zeros((m, n), np.float32)
    Bm = Bf.reshape(m, n)
    for i in range(0, m, 97):  # sampled rows: the serial fp32 FMA chain of csrmm_row_major_ref
        acc = np.zeros(n, np.float32)
        for p in range(rp[i] - base, rp[i + 1] - base):
            acc = (np.float64(vf[p]) * Bm[ci[p] - base].astype(np.float64) + acc.astype(np.float64)).astype(np.float32)
        assert np.array_equal(Cf.cpu().numpy().reshape(m, n)[i], acc), i


@pytest.mark.parametrize("base", [0, 1])
def test_csrmm_tiled_narrow_row_major_bit_exact(base):
    """csrmm_tile_kernel (row-major, n < 128: workgroup per row block of the SpMV plan, A staged in LDS): empty rows,
    rows longer than the LDS tile (a block of their own), padded leading dimensions, alpha / beta classes incl. the
    beta = 0 store path, NaN already in C (propagated by default as in the reference, overwritten in the opt-in mode), n from 2 to 126."""
    m, k = 2600, 2100
    rng = np.random.default_rng(17)
    def rowlen(r, i):
        if i in (5, 1300):
            return 1500  # longer than a 512- or 1024-entry tile
        return 0 if i % 97 == 3 else int(r.integers(1, 30))
    rp, ci, v = random_csr(23, m, k, rowlen, base=base)
    A = P.Matrix(base, m, k, rp, ci, v)
    d = P.Descr(base=base)
    assert L.aoclsparse_set_mm_hint(A.h, P.OP_NONE, d.h, 2) == 0 and L.aoclsparse_optimize(A.h) == 0
    for n, alpha, beta in ((2, 1.0, 0.0), (32, 1.0, 0.0), (32, -0.5, 2.0), (64, 3.0, -1.0), (96, 1.0, 0.0), (126, 0.25, 1.0)):
        ldb, ldc = n + 2, n + 4
        Br, C0 = rng.uniform(-1, 1, k * ldb), rng.uniform(-1, 1, m * ldc)
        Cd = dev(C0)
        assert P.dcsrmm(P.OP_NONE, alpha, A, d, P.ORDER_ROW, dev(Br), n, ldb, beta, Cd, ldc) == 0
        torch.cuda.synchronize()
        got = Cd.cpu().numpy().reshape(m, ldc)
        ref = _col_reference(alpha, base, v, ci, rp, m, k, Br, n, ldb, beta, C0, ldc)
        assert np.array_equal(got[:, :n], ref), "n=%d" % n
        assert np.array_equal(got[:, n:], C0.reshape(m, ldc)[:, n:])  # padding untouched
    # beta = 0 never reads a finite-or-not C
    n = 32
    Br = rng.uniform(-1, 1, k * n)
    Cd = torch.full((m * n,), float("nan"), dtype=torch.float64, device="cuda")
    with beta0_overwrite(P):
        assert P.dcsrmm(P.OP_NONE, 1.0, A, d, P.ORDER_ROW, dev(Br), n, n, 0.0, Cd, n) == 0
        torch.cuda.synchronize()
    ref = _col_reference(1.0, base, v, ci, rp, m, k, Br, n, n, 0.0, np.zeros(m * n), n)
    got = Cd.cpu().numpy().reshape(m, n)
    nonempty = np.diff(rp) > 0
    assert np.array_equal(got[nonempty], ref[nonempty])
    Cd = torch.full((m * n,), float("nan"), dtype=torch.float64, device="cuda")   # default: the reference's 0 * NaN
    assert P.dcsrmm(P.OP_NONE, 1.0, A, d, P.ORDER_ROW, dev(Br), n, n, 0.0, Cd, n) == 0
    torch.cuda.synchronize()
    assert bool(torch.isnan(Cd).all())


@pytest.mark.parametrize("base", [0, 1])
def test_csrmm_column_major_row_pairs_bit_exact(base):
    """csrmm_colpair_kernel: banded matrix whose odd rows carry the even rows' pattern shifted by one column (5-point
    Laplacian with perturbed values), odd row count, boundary rows that do not pair, a few rows made irregular on
    purpose: column-major C must equal csrmm_col_major_ref bit for bit; padded leading dimensions, beta classes."""
    g = 61  # odd grid edge: row pairs straddle grid lines, m = 3721 is odd
    m, rp, ci, v = laplace5(g, base=base)
    rng = np.random.default_rng(9)
    v = v * rng.uniform(0.5, 1.5, len(v))
    A = P.Matrix(base, m, m, rp, ci, v)
    d = P.Descr(base=base)
    assert L.aoclsparse_set_mm_hint(A.h, P.OP_NONE, d.h, 2) == 0 and L.aoclsparse_optimize(A.h) == 0
    for n, alpha, beta in ((4, 1.0, 0.0), (7, 2.0, 0.0), (70, -1.0, 0.5), (130, 1.0, 0.0)):
        ldb, ldc = m + 3, m + 5  # odd ldc: 16-byte stores not allowed -> generic kernel; even below
        for ldc in (m + 5, m + 6):
            B, C0 = rng.uniform(-1, 1, ldb * n), rng.uniform(-1, 1, ldc * n)
            Cd = dev(C0)
            assert P.dcsrmm(P.OP_NONE, alpha, A, d, P.ORDER_COLUMN, dev(B), n, ldb, beta, Cd, ldc) == 0
            torch.cuda.synchronize()
            so, Cr = oracle.dcsrmm("col", alpha, base, v, ci, rp, m, B, n, ldb, beta, C0, ldc)
            assert np.array_equal(Cd.cpu().numpy(), Cr), "n=%d ldc=%d" % (n, ldc)
    # a matrix whose pairs do NOT match must not take the pair kernel and stays exact
    rp2, ci2, v2 = random_csr(4, 900, 800, lambda r, i: r.integers(0, 9), base=base)
    A2 = P.Matrix(base, 900, 800, rp2, ci2, v2)
    B, C0 = rng.uniform(-1, 1, 800 * 12), rng.uniform(-1, 1, 900 * 12)
    Cd = dev(C0)
    assert P.dcsrmm(P.OP_NONE, 1.0, A2, d, P.ORDER_COLUMN, dev(B), 12, 800, -1.0, Cd, 900) == 0
    torch.cuda.synchronize()
    so, Cr = oracle.dcsrmm("col", 1.0, base, v2, ci2, rp2, 900, B, 12, 800, -1.0, C0, 900)
    assert np.array_equal(Cd.cpu().numpy(), Cr)


def test_csrmm_full_config_both_layouts_and_slab():
    """config 4 at full size (1M x 1M Laplacian, 256 columns, beta = 0): row-major and column-major products agree
    element for element (one FMA chain per element), the 32-column slab of an 8-rank run equals the matching columns of
    the full product in both layouts, and 4 columns equal the oracle."""
    g, n = 1000, 256
    m, rp, ci, v = laplace5(g)
    A = P.Matrix(0, m, m, rp, ci, v)
    d = P.Descr()
    assert L.aoclsparse_set_mm_hint(A.h, P.OP_NONE, d.h, 2) == 0 and L.aoclsparse_optimize(A.h) == 0
    L.aoclsparse_mi355_set_pointer_mode(P.PTR_DEVICE)
    try:
        gen = torch.Generator(device="cuda")
        gen.manual_seed(777)
        Bc = torch.rand(n * m, dtype=torch.float64, device="cuda", generator=gen) * 2 - 1  # column-major, ld = m
        Br = Bc.reshape(n, m).t().contiguous().reshape(-1)  # the same matrix row-major, ld = n
        Cc, Cr = torch.zeros(n * m, dtype=torch.float64, device="cuda"), torch.zeros(n * m, dtype=torch.float64, device="cuda")
        assert P.dcsrmm(P.OP_NONE, 1.0, A, d, P.ORDER_COLUMN, Bc, n, m, 0.0, Cc, m) == 0
        assert P.dcsrmm(P.OP_NONE, 1.0, A, d, P.ORDER_ROW, Br, n, n, 0.0, Cr, n) == 0
        torch.cuda.synchronize()
        assert torch.equal(Cc.reshape(n, m), Cr.reshape(m, n).t())
        j0, j1 = P.column_shard(n, 8, 5)
        assert (j0, j1) == (160, 192)
        w = j1 - j0
        Cs = torch.zeros(w * m, dtype=torch.float64, device="cuda")
        assert P.dcsrmm(P.OP_NONE, 1.0, A, d, P.ORDER_COLUMN, Bc[j0 * m:j1 * m], w, m, 0.0, Cs, m) == 0
        torch.cuda.synchronize()
        assert torch.equal(Cs, Cc[j0 * m:j1 * m])
        Bs = Br.reshape(m, n)[:, j0:j1].contiguous().reshape(-1)  # the slab a rank owns: m x 32, ld = 32 (tiled kernel)
        Cs.zero_()
        assert P.dcsrmm(P.OP_NONE, 1.0, A, d, P.ORDER_ROW, Bs, w, w, 0.0, Cs, w) == 0
        torch.cuda.synchronize()
        assert torch.equal(Cs.reshape(m, w), Cr.reshape(m, n)[:, j0:j1])
        for j in (0, 100, 161, 255):
            so, cr = oracle.dcsrmm("col", 1.0, 0, v, ci, rp, m, Bc[j * m:(j + 1) * m].cpu().numpy(), 1, m, 0.0, np.zeros(m), m)
            assert np.array_equal(Cc[j * m:(j + 1) * m].cpu().numpy(), cr)
    finally:
        L.aoclsparse_mi355_set_pointer_mode(P.PTR_AUTO)


# --------------------------------------------------------------------------------------------------
# per-iteration timing (aoclsparse_mi355_timer_mark / _laps)
# --------------------------------------------------------------------------------------------------
def test_timer_marks_give_one_lap_per_interval():
    m, rp, ci, v = laplace5(300)
    A = P.Matrix(0, m, m, rp, ci, v)
    d = P.Descr()
    x, y = dev(np.ones(m)), dev(np.zeros(m))
    assert P.timer_laps() == []  # nothing recorded: empty, not an error
    P.timer_mark()
    assert P.timer_laps() == []  # one mark, no interval
    P.timer_mark()
    for _ in range(7):
        assert P.dmv(P.OP_NONE, 1.0, A, d, x, 0.0, y) == 0
        P.timer_mark()
    laps = P.timer_laps()
    assert len(laps) == 7 and all(0.0 < t < 50.0 for t in laps)
    assert P.timer_laps() == []  # the ring was reset


# --------------------------------------------------------------------------------------------------
# TRSV robustness: NaN / Inf in b (mv_tests.cpp:1858-2290 style extreme values, applied to the solves) and a value
# whose bits equal the sync-free kernels' NOT-READY tag
# --------------------------------------------------------------------------------------------------
def _same_up_to_nan_payload(got, ref):
    """NaN where the reference has NaN (payloads may differ between x86 and gfx950), identical bits elsewhere"""
    gn, rn = np.isnan(got), np.isnan(ref)
    return bool(np.array_equal(gn, rn) and np.array_equal(got[~gn], ref[~rn]))


@pytest.mark.parametrize("fill,unit", [("lower", False), ("upper", True)])
def test_trsv_nan_inf_and_tag_collision_every_schedule(fill, unit):
    """NaN, +-Inf and the exact NOT-READY bit pattern (0x7FF8DEADBEEF0355, a NaN) placed in b must propagate through
    the dependency DAG exactly as in the serial reference chain -- for the per-level launches (schedule 0), the hybrid
    schedule (1), the sync-free kernels (2: lane per position, also what trsm runs; 3: slice per wavefront; 4: lane per block)
    and the automatic choice -- and never hang or report an error: a result equal to the tag is
    published as a plain quiet NaN."""
    from util import triangular_system
    m = 20000
    rp, ci, v = triangular_system(77, m, 4, band=300)
    A = P.Matrix(0, m, m, rp, ci, v)
    d = P.Descr(mtype=P.TYPE_TRIANGULAR, fill=P.FILL_LOWER if fill == "lower" else P.FILL_UPPER,
                diag=P.DIAG_UNIT if unit else P.DIAG_NON_UNIT)
    rng = np.random.default_rng(3)
    tag = np.array([0x7FF8DEADBEEF0355], dtype=np.uint64).view(np.float64)[0]
    o = oracle.dcsr_optimize(m, m, len(v), 0, rp, ci, v)
    kind, ilend = ("l", o["idiag"]) if fill == "lower" else ("u", o["iurow"])
    for case in ("nan", "inf", "tag", "mixed"):
        b = rng.uniform(-1, 1, m)
        if case in ("nan", "mixed"):
            b[[17, 9000]] = np.nan
        if case in ("inf", "mixed"):
            b[[300, 15000]] = [np.inf, -np.inf]
        if case in ("tag", "mixed"):
            b[[5, 12345]] = tag
        st, xr = oracle.dtrsv(kind, 1.0, m, 0, o["val"], o["ind"], o["ptr"], ilend, b, unit)
        assert st == 0
        assert np.isnan(xr).sum() >= 2 or case == "inf"
        for sched in (0, 1, 2, 3, 4, -1):  # every schedule, then the automatic choice
            with trsv_schedule(P, sched):
                xd = torch.zeros(m, dtype=torch.float64, device="cuda")
                assert P.dtrsv(P.OP_NONE, 1.0, A, d, dev(b), xd) == 0, (case, sched)
                torch.cuda.synchronize()
                assert _same_up_to_nan_payload(xd.cpu().numpy(), xr), (case, sched)
                xh = np.zeros(m)
                assert P.dtrsv(P.OP_NONE, 1.0, A, d, b, xh) == 0  # host pointers: synchronous, status checked
                assert _same_up_to_nan_payload(xh, xr), (case, sched)
        # the KT orders (kid 1 / 3) propagate NaN / Inf / the tag through the same DAG: NaN exactly where the chain has NaN
        for kid in (1, 3):
            xd = torch.zeros(m, dtype=torch.float64, device="cuda")
            assert P.dtrsv(P.OP_NONE, 1.0, A, d, dev(b), xd, kid=kid) == 0, (case, kid)
            torch.cuda.synchronize()
            assert np.array_equal(np.isnan(xd.cpu().numpy()), np.isnan(xr)), (case, kid)
        # two right-hand sides through trsm: the lane-per-position sync-free kernel
        Bm = np.stack([b, b[::-1].copy()])  # column-major, ld = m
        Xm = np.zeros_like(Bm)
        assert L.aoclsparse_dtrsm(P.OP_NONE, 1.0, A.h, d.h, P.ORDER_COLUMN, P._ptr(Bm), 2, m, P._ptr(Xm), m) == 0
        assert _same_up_to_nan_payload(Xm[0], xr), case
        st, xr2 = oracle.dtrsv(kind, 1.0, m, 0, o["val"], o["ind"], o["ptr"], ilend, Bm[1], unit)
        assert _same_up_to_nan_payload(Xm[1], xr2), case


def test_strsv_tag_collision_float():
    from util import triangular_system
    m = 5000
    rp, ci, v = triangular_system(78, m, 3, band=100, dtype=np.float32)
    A = P.Matrix(0, m, m, rp, ci, v)
    d = P.Descr(mtype=P.TYPE_TRIANGULAR, fill=P.FILL_LOWER)
    b = np.random.default_rng(4).uniform(-1, 1, m).astype(np.float32)
    b[[3, 2500]] = np.array([0x7FC0D355], dtype=np.uint32).view(np.float32)[0]
    for sched in (0, 2):  # per-level launches, sync-free
        with trsv_schedule(P, sched):
            xd = torch.zeros(m, dtype=torch.float32, device="cuda")
            assert P.strsv(P.OP_NONE, 1.0, A, d, dev(b), xd) == 0
            torch.cuda.synchronize()
            got = xd.cpu().numpy()
            if sched == 0:
                ref = got
    assert _same_up_to_nan_payload(got, ref) and np.isnan(ref).sum() >= 2


def test_trsv_sync_free_is_asynchronous_for_device_pointers():
    """device-pointer solves return before the kernel has finished (no stream sync, no blocking read of a timeout
    word): many solves can be queued back to back and the stream's own order keeps them correct."""
    import time
    from util import triangular_system
    m = 200000
    rp, ci, v = triangular_system(79, m, 3, band=50)
    A = P.Matrix(0, m, m, rp, ci, v)
    d = P.Descr(mtype=P.TYPE_TRIANGULAR, fill=P.FILL_LOWER)
    b = np.random.default_rng(5).uniform(-1, 1, m)
    bd, xd = dev(b), torch.zeros(m, dtype=torch.float64, device="cuda")
    assert P.dtrsv(P.OP_NONE, 1.0, A, d, bd, xd) == 0  # analysis + first solve
    torch.cuda.synchronize()
    lv = A.trsv_levels(P.FILL_LOWER)
    t0 = time.perf_counter()
    for _ in range(20):
        assert P.dtrsv(P.OP_NONE, 1.0, A, d, bd, xd) == 0
    t_enqueue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_total = time.perf_counter() - t0
    o = oracle.dcsr_optimize(m, m, len(v), 0, rp, ci, v)
    st, xr = oracle.dtrsv("l", 1.0, m, 0, o["val"], o["ind"], o["ptr"], o["idiag"], b, False)
    assert np.array_equal(xd.cpu().numpy(), xr)
    assert lv > 100 and t_enqueue < 0.5 * t_total, (t_enqueue, t_total)


# --------------------------------------------------------------------------------------------------
# raw aoclsparse_dcsrmv on device arrays: the implicit plan cache must never serve another matrix's plan
# --------------------------------------------------------------------------------------------------
def test_raw_dcsrmv_device_arrays_plan_cache_is_validated():
    """Two different matrices with the same m and nnz written one after the other into the SAME device buffers (what a
    caching allocator does after free + malloc): the second product must be the second matrix's, bit for bit -- the
    cached row-block plan of the first is checked against the live row_ptr and rebuilt."""
    m, n = 40000, 40000
    rng = np.random.default_rng(12)
    # matrix 1: every row 10 entries; matrix 2: same nnz, very different row lengths (blocks of the plan move)
    lens1 = np.full(m, 10, np.int64)
    lens2 = np.zeros(m, np.int64)
    lens2[: m // 2] = 1
    lens2[m // 2:] = 19
    mats = []
    for lens in (lens1, lens2):
        rp = np.zeros(m + 1, np.int32)
        rp[1:] = np.cumsum(lens)
        ci = np.concatenate([np.sort(rng.choice(n, size=int(k), replace=False)) for k in lens]).astype(np.int32)
        mats.append((rp, ci, rng.uniform(-1, 1, len(ci))))
    assert len(mats[0][1]) == len(mats[1][1])
    nnz = len(mats[0][1])
    d = P.Descr()
    x = rng.uniform(-1, 1, n)
    xd = dev(x)
    d_rp, d_ci, d_v = dev(mats[0][0]), dev(mats[0][1]), dev(mats[0][2])
    L.aoclsparse_mi355_set_pointer_mode(P.PTR_DEVICE)
    try:
        for rep in range(2):
            for rp, ci, v in mats:
                d_rp.copy_(torch.from_numpy(rp)), d_ci.copy_(torch.from_numpy(ci)), d_v.copy_(torch.from_numpy(v))
                torch.cuda.synchronize()
                yd = torch.zeros(m, dtype=torch.float64, device="cuda")
                for _ in range(2):  # second call: a validated cache hit
                    assert P.dcsrmv(P.OP_NONE, 1.0, m, n, nnz, d_v, d_ci, d_rp, d, xd, 0.0, yd) == 0
                torch.cuda.synchronize()
                so, yr = oracle.dcsrmv(-1, 0, 1.0, m, nnz, v, ci, rp, x, 0.0, np.zeros(m))
                assert np.array_equal(yd.cpu().numpy(), yr)
    finally:
        L.aoclsparse_mi355_set_pointer_mode(P.PTR_AUTO)


# --------------------------------------------------------------------------------------------------
# reference vector: triangular aoclsparse_?mv on a rectangular matrix (mv_tests.cpp:344-388)
# --------------------------------------------------------------------------------------------------
def test_mv_triangular_rectangular_reference_kat(kats):
    k = kats["mv_tri"]
    m, n = k["m"], k["n"]
    for base in (0, 1):
        rp, ci = np.array(k["row_ptr"], np.int32) + base, np.array(k["col_ind"], np.int32) + base
        x = np.array(k["x"], np.float64)
        for dtype, mvf in ((np.float64, P.dmv), (np.float32, P.smv)):
            v = np.array(k["val"], dtype)
            for fill, gold in ((P.FILL_LOWER, k["exp_y_l"]), (P.FILL_UPPER, k["exp_y_u"])):
                A = P.Matrix(base, m, n, rp, ci, v)  # a fresh handle per fill mode, as the reference test does
                d = P.Descr(base=base, mtype=P.TYPE_TRIANGULAR, fill=fill)
                for on_device in (False, True):
                    y = np.full(m, np.nan, dtype)  # beta = 0: y is overwritten, never read
                    xx = x.astype(dtype)
                    if on_device:
                        yd = dev(y)
                        assert mvf(P.OP_NONE, k["alpha"], A, d, dev(xx), k["beta"], yd) == 0
                        torch.cuda.synchronize()
                        y = yd.cpu().numpy()
                    else:
                        assert mvf(P.OP_NONE, k["alpha"], A, d, xx, k["beta"], y) == 0
                    assert np.array_equal(y, np.array(gold, dtype)), (base, dtype, fill, on_device)


# --------------------------------------------------------------------------------------------------
# the third kernel aoclsparse_optimize can choose: merge-path for matrices with very long rows
# --------------------------------------------------------------------------------------------------
def test_device_pointer_calls_can_be_captured_in_a_hip_graph():
    """device-pointer ?mv / ?trsv calls enqueue kernels and memsets only (no allocation, no synchronisation, no host
    read-back once their workspaces exist), so a solver loop can be captured on the stream handed to
    aoclsparse_mi355_set_stream and replayed as one HIP graph; results are those of the eager calls, bit for bit"""
    import ctypes
    from test_gpu_trsv_blocks import node_mesh
    nodes = 2000
    m, rp, ci, v = node_mesh(9, nodes, 40, np.full(nodes, 5))
    A = P.Matrix(0, m, m, rp, ci, v)
    dg = P.Descr()
    dl = P.Descr(mtype=P.TYPE_TRIANGULAR, fill=P.FILL_LOWER)
    du = P.Descr(mtype=P.TYPE_TRIANGULAR, fill=P.FILL_UPPER)
    L.aoclsparse_mi355_set_pointer_mode(P.PTR_DEVICE)
    s = torch.cuda.Stream()
    assert L.aoclsparse_mi355_set_stream(ctypes.c_void_p(s.cuda_stream)) == 0
    try:
        with torch.cuda.stream(s):
            x = dev(np.random.default_rng(3).uniform(-1, 1, m))
            y, z, w = (torch.zeros(m, dtype=torch.float64, device="cuda") for _ in range(3))

            def step():
                assert P.dmv(P.OP_NONE, 1.0, A, dg, x, 0.0, y) == 0          # y = A x
                assert P.dtrsv(P.OP_NONE, 1.0, A, dl, y, z) == 0            # z = L^-1 y
                assert P.dtrsv(P.OP_NONE, 1.0, A, du, z, w) == 0            # w = U^-1 z

            for _ in range(2):  # workspaces, plans
                step()
            s.synchronize()
            ref = w.clone()
            w.zero_(), z.zero_(), y.zero_()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                step()
            for _ in range(3):
                g.replay()
            s.synchronize()
            assert torch.equal(w, ref)
    finally:
        assert L.aoclsparse_mi355_set_stream(None) == 0
        L.aoclsparse_mi355_set_pointer_mode(P.PTR_AUTO)


def test_heavy_first_block_order_is_bit_identical():
    """row blocks that hold a long row are handed to the first workgroups (SpmvPlan::rowblocks4): same blocks, same
    per-row chains -- the product must stay bit-identical to the serial scalar-order reference, through a handle and
    through the raw-array entry, for base 0 and 1 (strict mode: every row a chain; the automatic mode's rows of fewer than 32
    entries are compared too)"""
    from util import powerlaw_rows
    m = 120000
    for base in (0, 1):
        rp, ci, v = random_csr(91 + base, m, m, powerlaw_rows(6, 400), base=base)
        assert len(v) > 512 * 600 and np.diff(rp).max() >= 64  # enough blocks, heavy rows present
        x = np.random.default_rng(5).uniform(-1, 1, m)
        y0 = np.random.default_rng(6).uniform(-1, 1, m)
        st, yr = oracle.dcsrmv(0, base, 1.5, m, len(v), v, ci, rp, x, -0.5, y0)
        assert st == 0
        A = P.Matrix(base, m, m, rp, ci, v)
        d = P.Descr(base=base)
        assert L.aoclsparse_set_mv_hint(A.h, P.OP_NONE, d.h, 100) == 0 and L.aoclsparse_optimize(A.h) == 0
        assert A.spmv_info().kernel == 1  # csr-adaptive (not SELL, not merge-path)
        short = np.diff(rp) < 32
        yd = dev(y0)
        assert P.dmv(P.OP_NONE, 1.5, A, d, dev(x), -0.5, yd) == 0
        torch.cuda.synchronize()
        assert np.array_equal(yd.cpu().numpy()[short], yr[short]), base
        assert L.aoclsparse_mi355_set_option(P.OPTION_SPMV_STRICT, 1) == 0
        try:
            yd = dev(y0)
            assert P.dmv(P.OP_NONE, 1.5, A, d, dev(x), -0.5, yd) == 0
            torch.cuda.synchronize()
            assert np.array_equal(yd.cpu().numpy(), yr), base
            yh = y0.copy()
            assert P.dcsrmv(P.OP_NONE, 1.5, m, m, len(v), v, ci, rp, d, x, -0.5, yh) == 0
            assert np.array_equal(yh, yr), base
        finally:
            assert L.aoclsparse_mi355_set_option(P.OPTION_SPMV_STRICT, 0) == 0


def test_optimize_selects_merge_path_for_very_long_rows():
    """Row-length statistics decide the SpMV kernel: a tridiagonal matrix with two rows of ~45,000 entries (> 32 LDS
    tiles) gets merge-path tiles, the same matrix with rows of ~4,000 entries stays on CSR-Adaptive.  Rows inside one
    tile are bit-exact (scalar order); a row cut by tile boundaries is within (len + pieces + 8) eps sum|a x|."""
    n = 120000
    rng = np.random.default_rng(21)
    tri = np.stack([np.arange(n) - 1, np.arange(n), np.arange(n) + 1], axis=1)
    for length, want in ((60000, 2), (4000, 1)):
        longs = {777, 90001}
        parts, rp = [], np.zeros(n + 1, np.int64)
        for i in range(n):
            c = np.unique(np.concatenate([rng.integers(0, n, size=length), [i]])) if i in longs else tri[i][(tri[i] >= 0) & (tri[i] < n)]
            parts.append(c)
            rp[i + 1] = rp[i] + len(c)
        ci, rp = np.concatenate(parts).astype(np.int32), rp.astype(np.int32)
        v = rng.uniform(-1, 1, len(ci))
        A = P.Matrix(0, n, n, rp, ci, v)
        d = P.Descr()
        assert L.aoclsparse_mi355_set_option(P.OPTION_SELL, 0) == 0  # no SELL copy: the choice between the two CSR kernels
        try:
            assert L.aoclsparse_set_mv_hint(A.h, P.OP_NONE, d.h, 100) == 0 and L.aoclsparse_optimize(A.h) == 0
        finally:
            assert L.aoclsparse_mi355_set_option(P.OPTION_SELL, -1) == 0
        info = A.spmv_info()
        assert info.kernel == want, (length, info.kernel, info.max_row_nnz, info.tile)
        x = rng.uniform(-1, 1, n)
        yd = torch.zeros(n, dtype=torch.float64, device="cuda")
        assert P.dmv(P.OP_NONE, 1.0, A, d, dev(x), 0.0, yd) == 0
        torch.cuda.synchronize()
        so, yr = oracle.dcsrmv(0, 0, 1.0, n, len(v), v, ci, rp, x, 0.0, np.zeros(n))
        got, lens = yd.cpu().numpy(), np.diff(rp)
        if want == 1:
            assert np.array_equal(got[lens <= 3], yr[lens <= 3])  # row-block kernel: every row inside a tile is exact
        else:
            # merge-path: a row is exact unless one of the 1,024-item tile boundaries cuts it (at most one row per tile)
            ntiles = (n + len(v)) // 1024 + 1
            assert int(np.sum(got != yr)) <= ntiles + 2
        scale = np.add.reduceat(np.abs(v * x[ci]), rp[:-1])
        bound = (lens + lens / 512.0 + 8 + 2 * np.ceil(np.log2(np.maximum(lens, 2)))) * EPS64 * scale
        assert np.all(np.abs(got - yr) <= bound + 1e-300)


# --------------------------------------------------------------------------------------------------
# host-pointer calls with large arrays (tens of MB per operand, staged per call)
# --------------------------------------------------------------------------------------------------
def test_large_host_pointer_calls():
    """raw aoclsparse_dcsrmv with ALL arrays on the host (three transfers of 21-42 MB back to back), then aoclsparse_dmv
    with host x / y on a handle, then a host-pointer csrmm: results must be the oracle's bits (the staging buffers are
    reused from call to call, the copies are stream-ordered)."""
    g = 1024
    m, rp, ci, v = laplace5(g)
    rng = np.random.default_rng(31)
    v = v * rng.uniform(0.5, 1.5, len(v))
    nnz = len(v)
    x = rng.uniform(-1, 1, m)
    d = P.Descr()
    so, yr = oracle.dcsrmv(-1, 0, 1.0, m, nnz, v, ci, rp, x, 0.0, np.zeros(m), nthreads=4)
    for rep in range(3):
        y = np.full(m, np.nan)
        assert P.dcsrmv(P.OP_NONE, 1.0, m, m, nnz, v, ci, rp, d, x, 0.0, y) == 0
        assert np.array_equal(y, yr), "raw host-array csrmv, repetition %d" % rep
    A = P.Matrix(0, m, m, rp, ci, v)
    assert L.aoclsparse_set_mv_hint(A.h, P.OP_NONE, d.h, 10) == 0 and L.aoclsparse_optimize(A.h) == 0
    y = np.full(m, np.nan)
    assert P.dmv(P.OP_NONE, 1.0, A, d, x, 0.0, y) == 0
    assert np.array_equal(y, yr)
    n = 8  # csrmm, host B and C: 8.4 MB each way per operand
    B, C0 = rng.uniform(-1, 1, m * n), rng.uniform(-1, 1, m * n)
    C = C0.copy()
    assert P.dcsrmm(P.OP_NONE, 2.0, A, d, P.ORDER_COLUMN, B, n, m, -1.0, C, m) == 0
    so, Cr = oracle.dcsrmm("col", 2.0, 0, v, ci, rp, m, B, n, m, -1.0, C0, m)
    assert np.array_equal(C, Cr)


# --------------------------------------------------------------------------------------------------
# csrmm row groups on mesh matrices (the super-group kernel of round 2, which lost to them, was removed in round 3)
# --------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("which", ["shell", "flan"])
def test_csrmm_row_groups_on_mesh_matrices_bit_exact(which):
    """csrmm_rowgroup2_kernel (row-major, n >= 128): block-structured mesh matrices (5 dofs / 7 neighbours and 3 dofs /
    27 neighbours, small instances of the config-3 stand-ins).  Per element the chain is still the row in CSR order:
    C equals csrmm_col_major_ref's bits; a NaN / Inf in one B row reaches exactly the rows that reference that column;
    values changed with aoclsparse_dupdate_values are picked up."""
    if which == "shell":
        m, rp, ci, v = standins.shell_like(n=5 * 41 * 23, width=41)
    else:
        m, rp, ci, v = standins.flan_like(nx=9, ny=8, nz=7)
    rng = np.random.default_rng(33)
    _super_group_checks(which, m, rp, ci, v, rng)


def _super_group_checks(which, m, rp, ci, v, rng):
    A = P.Matrix(0, m, m, rp, ci, v)
    d = P.Descr()
    assert L.aoclsparse_set_mm_hint(A.h, P.OP_NONE, d.h, 2) == 0 and L.aoclsparse_optimize(A.h) == 0
    for n, alpha, beta in ((128, 1.0, 0.0), (256, -0.5, 1.5), (130, 2.0, 0.0)):
        ldb, ldc = n + 2, n + 4
        Br, C0 = rng.uniform(-1, 1, m * ldb), rng.uniform(-1, 1, m * ldc)
        Cd = dev(C0)
        assert P.dcsrmm(P.OP_NONE, alpha, A, d, P.ORDER_ROW, dev(Br), n, ldb, beta, Cd, ldc) == 0
        torch.cuda.synchronize()
        info = A.spmv_info()
        assert info.mm_groups > 0
        got = Cd.cpu().numpy().reshape(m, ldc)
        ref = _col_reference(alpha, 0, v, ci, rp, m, m, Br, n, ldb, beta, C0, ldc)
        assert np.array_equal(got[:, :n], ref), (which, n)
        assert np.array_equal(got[:, n:], C0.reshape(m, ldc)[:, n:])
    # NaN / Inf in B rows c1, c2: only rows that have those columns may see them
    n = 128
    Br = rng.uniform(-1, 1, m * n)
    c1, c2 = int(ci[rp[m // 2]]), int(ci[rp[m // 3 + 1] - 1])
    Br.reshape(m, n)[c1, 5] = np.nan
    Br.reshape(m, n)[c2, 77] = np.inf
    Cd = torch.zeros(m * n, dtype=torch.float64, device="cuda")
    assert P.dcsrmm(P.OP_NONE, 1.0, A, d, P.ORDER_ROW, dev(Br), n, n, 0.0, Cd, n) == 0
    torch.cuda.synchronize()
    got = Cd.cpu().numpy().reshape(m, n)
    ref = _col_reference(1.0, 0, v, ci, rp, m, m, Br, n, n, 0.0, np.zeros(m * n), n)
    bad_g, bad_r = ~np.isfinite(got), ~np.isfinite(ref)
    assert np.array_equal(bad_g, bad_r) and np.array_equal(got[~bad_g], ref[~bad_r])
    rows_with_c1 = {i for i in range(m) if c1 in ci[rp[i]:rp[i + 1]]}
    assert set(np.nonzero(np.isnan(got[:, 5]))[0]) == rows_with_c1
    # values updated in place: the dense block copy must follow
    v2 = v * rng.uniform(0.5, 1.5, len(v))
    assert L.aoclsparse_dupdate_values(A.h, len(v2), P._ptr(v2)) == 0
    Br = rng.uniform(-1, 1, m * n)
    Cd = torch.zeros(m * n, dtype=torch.float64, device="cuda")
    assert P.dcsrmm(P.OP_NONE, 1.0, A, d, P.ORDER_ROW, dev(Br), n, n, 0.0, Cd, n) == 0
    torch.cuda.synchronize()
    ref = _col_reference(1.0, 0, v2, ci, rp, m, m, Br, n, n, 0.0, np.zeros(m * n), n)
    assert np.array_equal(Cd.cpu().numpy().reshape(m, n), ref)
