#!/usr/bin/env python3
"""Secondary measurements (not the driver's bench line): SuiteSparse stand-ins (BASELINE config 3),
csrmm layouts (config 4), level-scheduled TRSV on ILU(0) factors (config 5), and the PCIe-inclusive
host-pointer SpMV rate.  Every result is checked against the CPU oracle.  One JSON object per line."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as entry  # noqa: E402
import oracle  # noqa: E402
import standins  # noqa: E402
from bench import csrmm_bytes, spmv_bytes  # noqa: E402

pkg = entry.load_package()
L = pkg.lib()
ap = argparse.ArgumentParser()
ap.add_argument("--what", default="spmv,csrmm,trsv,cg,next,wider,setup,pcie")
ap.add_argument("--small", action="store_true", help="skip the two 50-120 M nnz stand-ins")
args = ap.parse_args()
what = set(args.what.split(","))
dev = torch.device("cuda", 0)


def emit(**kw):
    print(json.dumps(kw), flush=True)


def time_calls(fn, reps, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    pkg.timer_start()
    for _ in range(reps):
        fn()
    return pkg.timer_stop() / reps


d0 = pkg.Descr()

if "spmv" in what:
    L.aoclsparse_mi355_set_pointer_mode(pkg.PTR_DEVICE)
    names = ["circuit-like", "web-like"] + ([] if args.small else ["shell-like", "flan-like"])
    for name in names:
        label, m, rp, ci, v = standins.load(name)
        nnz = len(v)
        A = pkg.Matrix(0, m, m, rp, ci, v)
        assert L.aoclsparse_set_mv_hint(A.h, pkg.OP_NONE, d0.h, 100) == 0 and L.aoclsparse_optimize(A.h) == 0
        info = A.spmv_info()
        xh = np.random.default_rng(1).uniform(-1, 1, m)
        x = torch.from_numpy(xh).to(dev)
        y = torch.zeros(m, dtype=torch.float64, device=dev)
        ms = time_calls(lambda: pkg.dmv(pkg.OP_NONE, 1.0, A, d0, x, 0.0, y), 50)
        so, yr = oracle.dcsrmv(-1, 0, 1.0, m, nnz, v, ci, rp, xh, 0.0, np.zeros(m), nthreads=oracle.max_threads())
        yd = y.cpu().numpy()
        lens = np.diff(rp)
        short = lens <= info.tile
        b = spmv_bytes(m, m, nnz)
        emit(kind="spmv", matrix=label, m=m, nnz=nnz, kernel={1: "csr-adaptive", 3: "sell-64"}.get(info.kernel, info.kernel),
             cells_per_nnz=round(info.stored_cells / nnz, 3) if info.kernel == 3 else None, order=info.order, tile=info.tile,
             row_blocks=info.row_blocks, long_rows=info.long_rows, max_row=int(lens.max()), ms=round(ms, 5),
             gflops=round(2 * nnz / ms / 1e6, 2), gbs=round(b / ms / 1e6, 1), frac_of_8TBs=round(b / ms / 1e6 / 8000, 4),
             bit_exact_rows_within_tile=bool(np.array_equal(yd[short], yr[short])),
             max_abs_diff_long_rows=float(np.max(np.abs(yd - yr))))
        del A, x, y

if "csrmm" in what:
    L.aoclsparse_mi355_set_pointer_mode(pkg.PTR_DEVICE)
    g = 1000
    m, rp, ci, v = entry.laplace5(g)
    nnz = len(v)
    A = pkg.Matrix(0, m, m, rp, ci, v)
    assert L.aoclsparse_set_mm_hint(A.h, pkg.OP_NONE, d0.h, 10) == 0 and L.aoclsparse_optimize(A.h) == 0
    for n in (256, 32):
        gen = torch.Generator(device=dev)
        gen.manual_seed(777)
        B = torch.rand(m * n, dtype=torch.float64, device=dev, generator=gen) * 2 - 1
        C = torch.zeros(m * n, dtype=torch.float64, device=dev)
        for order, ldb, ldc, nm in ((pkg.ORDER_ROW, n, n, "row-major"), (pkg.ORDER_COLUMN, m, m, "column-major")):
            for beta in (0.0, -2.0):
                ms = time_calls(lambda: pkg.dcsrmm(pkg.OP_NONE, 1.0, A, d0, order, B, n, ldb, beta, C, ldc), 10, 2)
                b = csrmm_bytes(m, m, nnz, n, beta != 0.0)
                emit(kind="csrmm", A="5-pt Laplacian grid 1000^2 (nnz=%d)" % nnz, n=n, layout=nm, beta=beta,
                     ms=round(ms, 4), gflops=round(2.0 * nnz * n / ms / 1e6, 1), gbs=round(b / ms / 1e6, 1),
                     frac_of_8TBs=round(b / ms / 1e6 / 8000, 4))
        # parity sample: 64 columns' worth against the oracle (col-major, beta = 0)
        C.zero_()
        pkg.dcsrmm(pkg.OP_NONE, 1.0, A, d0, pkg.ORDER_COLUMN, B, n, m, 0.0, C, m)
        torch.cuda.synchronize()
        ns = min(n, 8)
        so, Cr = oracle.dcsrmm("col", 1.0, 0, v, ci, rp, m, B[: ns * m].cpu().numpy(), ns, m, 0.0, np.zeros(ns * m), m)
        emit(kind="csrmm-parity", n=n, cols_checked=ns, bit_exact=bool(np.array_equal(C[: ns * m].cpu().numpy(), Cr)))
        del B, C

if "csrmm" in what and not args.small:
    # block-structured A (tens of non-zeros per row): the row-group kernel and the column-major detour
    L.aoclsparse_mi355_set_pointer_mode(pkg.PTR_DEVICE)
    for name, gen in (("shell-like (stand-in)", standins.shell_like), ("flan-like (stand-in)", standins.flan_like)):
        m, rp, ci, v = gen()
        nnz = len(v)
        A = pkg.Matrix(0, m, m, rp, ci, v)
        assert L.aoclsparse_set_mm_hint(A.h, pkg.OP_NONE, d0.h, 10) == 0 and L.aoclsparse_optimize(A.h) == 0
        for n in (256, 32):
            gen_t = torch.Generator(device=dev)
            gen_t.manual_seed(7)
            B = torch.rand(m * n, dtype=torch.float64, device=dev, generator=gen_t) * 2 - 1
            C = torch.zeros(m * n, dtype=torch.float64, device=dev)
            for order, ld, nm in ((pkg.ORDER_ROW, n, "row-major"), (pkg.ORDER_COLUMN, m, "column-major")):
                ms = time_calls(lambda: pkg.dcsrmm(pkg.OP_NONE, 1.0, A, d0, order, B, n, ld, 0.0, C, ld), 5, 2)
                b = csrmm_bytes(m, m, nnz, n, False)
                emit(kind="csrmm", A="%s m=%d nnz=%d" % (name, m, nnz), n=n, layout=nm, beta=0.0, ms=round(ms, 4),
                     row_groups=int(A.spmv_info().mm_groups), gflops=round(2.0 * nnz * n / ms / 1e6, 1),
                     gbs=round(b / ms / 1e6, 1), frac_of_8TBs=round(b / ms / 1e6 / 8000, 4))
            del B, C
        del A

if "trsv" in what:
    L.aoclsparse_mi355_set_pointer_mode(pkg.PTR_DEVICE)
    cases = [("ILU(0) of 5-pt Laplacian grid 1000^2", lambda: entry.laplace5(1000))]
    if not args.small:
        cases.append(("ILU(0) of shell-like stand-in (af_shell10-like)", standins.shell_like))
    for title, gen in cases:
        m, rp, ci, v = gen()
        t = time.time()
        st, lu, dg = oracle.dilu0(m, 0, rp, ci, v)
        t_ilu = time.time() - t
        assert st == 0, st
        A = pkg.Matrix(0, m, m, rp, ci, lu)
        dl = pkg.Descr(mtype=pkg.TYPE_TRIANGULAR, fill=pkg.FILL_LOWER, diag=pkg.DIAG_UNIT)
        t = time.time()
        assert L.aoclsparse_set_sv_hint(A.h, pkg.OP_NONE, dl.h, 100) == 0 and L.aoclsparse_optimize(A.h) == 0
        t_opt = time.time() - t
        lv = A.trsv_levels(pkg.FILL_LOWER)
        o = oracle.dcsr_optimize(m, m, len(lu), 0, rp, ci, lu)
        nnz_l = int(np.sum(o["idiag"] - rp[:-1]))
        bh = np.random.default_rng(2).uniform(-1, 1, m)
        t = time.time()
        st, xr = oracle.dtrsv("l", 1.0, m, 0, lu, ci, rp, o["idiag"], bh, True)
        t_cpu = time.time() - t
        bdev = torch.from_numpy(bh).to(dev)
        xdev = torch.zeros(m, dtype=torch.float64, device=dev)
        abytes = (m + 1 + nnz_l) * 4 + (2 * m + nnz_l) * 8
        for sched, nm in ((0, "one launch per level"), (1, "hybrid: narrow level runs in one workgroup"),
                          (2, "sync-free, lane per position"), (3, "sync-free, level slice per wavefront"),
                          (4, "sync-free, lane per block of chained rows (falls back to 3 / 2 without blocks)"), (-1, "automatic")):
            if sched in (0, 2, 3, 4) and lv > 100000:
                continue  # hundreds of thousands of launches / hops: minutes
            reps = 3 if sched != 1 and lv > 500 else 10
            assert L.aoclsparse_mi355_set_trsv_schedule(sched) == 0
            ms = time_calls(lambda: pkg.dtrsv(pkg.OP_NONE, 1.0, A, dl, bdev, xdev), reps, 1)
            assert L.aoclsparse_mi355_set_trsv_schedule(-1) == 0
            torch.cuda.synchronize()
            xg = xdev.cpu().numpy()
            emit(kind="trsv", system=title, m=m, nnz_strict_lower=nnz_l, levels=lv, schedule=nm, ms=round(ms, 4),
                 gflops=round((2.0 * nnz_l + m) / ms / 1e6, 2), gbs=round(abytes / ms / 1e6, 1),
                 cpu_serial_ms=round(t_cpu * 1e3, 2), bit_exact_vs_cpu=bool(np.array_equal(xg, xr)),
                 analysis_s=round(t_opt, 2), ilu0_cpu_s=round(t_ilu, 2))
        del A

if "cg" in what:
    # device-resident CG (aoclsparse_itsol_d_solve): every iterate stays in HBM; one host wait per iteration
    import ctypes
    import time
    L.aoclsparse_mi355_set_pointer_mode(pkg.PTR_AUTO)
    for g, pre in ((2048, "None"), (1024, "SymGS")):
        m, rp, ci, v = entry.laplace5(g)
        keep = ci <= np.repeat(np.arange(m, dtype=np.int32), np.diff(rp))
        lrp = np.concatenate([[0], np.cumsum(np.add.reduceat(keep.astype(np.int64), rp[:-1]))]).astype(np.int32)
        lci, lv = ci[keep].copy(), v[keep].copy()
        A = pkg.Matrix(0, m, m, lrp, lci, lv)
        ds = pkg.Descr(mtype=pkg.TYPE_SYMMETRIC, fill=pkg.FILL_LOWER)
        iters = 100
        h = ctypes.c_void_p()
        assert L.aoclsparse_itsol_d_init(ctypes.byref(h)) == 0
        for k, val in (("CG Iteration Limit", iters - 1), ("CG Rel Tolerance", 0.0), ("CG Abs Tolerance", 1e-300),
                       ("CG Preconditioner", pre)):
            assert L.aoclsparse_itsol_option_set(h, k.encode(), str(val).encode()) == 0
        xe = np.sin(0.001 * np.arange(m))
        so, b = oracle.dcsrmv(0, 0, 1.0, m, len(v), v, ci, rp, xe, 0.0, np.zeros(m), nthreads=oracle.max_threads())
        bd = torch.from_numpy(b).to(dev)
        rinfo = np.zeros(100)
        best = 1e30
        for rep in range(3):  # first pass pays for the analysis (symmetric expansion, level sets)
            xd = torch.zeros(m, dtype=torch.float64, device=dev)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            st = L.aoclsparse_itsol_d_solve(h, m, A.h, ds.h, pkg._ptr(bd), pkg._ptr(xd), pkg._ptr(rinfo), None, None, None)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        assert st == 7 and rinfo[30] == iters, (st, rinfo[30])
        o = oracle.dcsr_optimize(m, m, len(v), 0, rp, ci, v)
        t0 = time.perf_counter()
        so, xo, ro = oracle.dcg(m, 0, o["ptr"], o["ind"], o["val"], o["idiag"], o["iurow"], b, np.zeros(m), 0.0, 1e-300,
                                9, 3 if pre == "SymGS" else 0)
        cpu_ms_per_iter = (time.perf_counter() - t0) * 1e3 / max(ro[30], 1)
        emit(kind="cg", system="5-pt Laplacian grid %d^2 (m=%d), lower triangle stored, symmetric descriptor" % (g, m),
             preconditioner=pre, iterations=iters, ms_per_iteration=round(best * 1e3 / iters, 4),
             residual_after=float(rinfo[0]), cpu_serial_ms_per_iteration=round(cpu_ms_per_iter, 2),
             note="CPU = the restated reference loop, one thread (its level-1 steps are serial loops)")
        L.aoclsparse_itsol_destroy(ctypes.byref(h))
        del A

if "next" in what:
    # the SURVEY 8f rows, one line each: same device-resident setting as the headline, parity stated per line
    import ctypes
    L.aoclsparse_mi355_set_pointer_mode(pkg.PTR_DEVICE)
    g = 2048
    m, rp, ci, v = entry.laplace5(g)
    nnz = len(v)
    rng = np.random.default_rng(5)
    xh = rng.uniform(-1, 1, m)
    xd = torch.from_numpy(xh).to(dev)
    so, yref = oracle.dcsrmv(-1, 0, 1.0, m, nnz, v, ci, rp, xh, 0.0, np.zeros(m), nthreads=oracle.max_threads())
    one, zero = np.array([1.0]), np.array([0.0])
    # ELLT twin (device arrays)
    w, tc, tv = oracle.csr2ell("ellt", m, 0, rp, ci, v)
    tcd, tvd, yd = torch.from_numpy(tc).to(dev), torch.from_numpy(tv).to(dev), torch.zeros(m, dtype=torch.float64, device=dev)
    fn = lambda: L.aoclsparse_delltmv(pkg.OP_NONE, pkg._ptr(one), m, m, nnz, pkg._ptr(tvd), pkg._ptr(tcd), w, d0.h, pkg._ptr(xd),
                                     pkg._ptr(zero), pkg._ptr(yd))
    ms = time_calls(fn, 30)
    emit(kind="next", op="aoclsparse_delltmv (device arrays)", system="5-pt Laplacian grid %d^2" % g, ms=round(ms, 4),
         gbs_ell_bytes=round((m * w * 12 + 16 * m) / ms / 1e6, 1), bit_exact_vs_csr_scalar_order=bool(np.array_equal(yd.cpu().numpy(), yref)))
    # BLKCSR twin on the blocked stand-in (conversion by the library's own host routine, product on device arrays)
    import standins
    bm, brp_, bci_, bv_ = standins.shell_like()
    bnnz, tot = len(bv_), ctypes.c_int32(0)
    rows_blk = L.aoclsparse_opt_blksize(bm, bnnz, 0, pkg._ptr(brp_), pkg._ptr(bci_), ctypes.byref(tot)) or 4
    brp, bc = np.zeros(bm + 1, np.int32), np.zeros(bnnz, np.int32)
    bv, mk = np.zeros(bnnz + 64), np.zeros(bnnz * rows_blk + 64, np.uint8)
    assert L.aoclsparse_csr2blkcsr(bm, bm, bnnz, pkg._ptr(brp_), pkg._ptr(bci_), pkg._ptr(bv_), pkg._ptr(brp), pkg._ptr(bc), pkg._ptr(bv),
                                   pkg._ptr(mk), rows_blk, 0) == 0
    nb = int(brp[bm])
    bx = rng.uniform(-1, 1, bm)
    so, byref = oracle.dblkcsrmv(0, 1.0, bm, mk[: nb * rows_blk], bv[:bnnz], bc[:nb], brp, bx, 0.0, np.zeros(bm), rows_blk)
    t = lambda a: torch.from_numpy(a).to(dev)
    mkd, bvd, bcd, brpd, bxd, byd = t(mk), t(bv), t(bc), t(brp), t(bx), torch.zeros(bm, dtype=torch.float64, device=dev)
    fn = lambda: L.aoclsparse_dblkcsrmv(pkg.OP_NONE, pkg._ptr(one), bm, bm, bnnz, pkg._ptr(mkd), pkg._ptr(bvd), pkg._ptr(bcd),
                                        pkg._ptr(brpd), d0.h, pkg._ptr(bxd), pkg._ptr(zero), pkg._ptr(byd), rows_blk)
    ms = time_calls(fn, 30)
    emit(kind="next", op="aoclsparse_dblkcsrmv %dx8 (device arrays)" % rows_blk,
         system="shell-like stand-in m=%d nnz=%d blocks=%d" % (bm, bnnz, nb), ms=round(ms, 4),
         gbs_blk_bytes=round((bnnz * 8 + nb * (4 + rows_blk) + 4 * bm + 16 * bm) / ms / 1e6, 1),
         bit_exact_vs_avx512_order=bool(np.array_equal(byd.cpu().numpy(), byref)))
    # dotmv
    A = pkg.Matrix(0, m, m, rp, ci, v)
    assert L.aoclsparse_set_mv_hint(A.h, pkg.OP_NONE, d0.h, 10) == 0 and L.aoclsparse_optimize(A.h) == 0
    dd = torch.zeros(1, dtype=torch.float64, device=dev)
    ms = time_calls(lambda: L.aoclsparse_ddotmv(pkg.OP_NONE, 1.0, A.h, d0.h, pkg._ptr(xd), 0.0, pkg._ptr(yd), pkg._ptr(dd)), 30)
    emit(kind="next", op="aoclsparse_ddotmv", system="5-pt Laplacian grid %d^2" % g, ms=round(ms, 4),
         y_bit_exact=bool(np.array_equal(yd.cpu().numpy(), yref)), dot_rel_err=float(abs(dd.item() - np.dot(xh, yref)) / abs(np.dot(xh, yref))))
    # complex SpMV (same pattern, complex values)
    vz = (v + 1j * rng.uniform(-1, 1, nnz)).astype(np.complex128)
    hz = ctypes.c_void_p()
    assert L.aoclsparse_create_zcsr(ctypes.byref(hz), 0, m, m, nnz, pkg._ptr(rp), pkg._ptr(ci), pkg._ptr(vz)) == 0
    xz = torch.from_numpy((xh + 1j * rng.uniform(-1, 1, m)).astype(np.complex128)).to(dev)
    yz = torch.zeros(m, dtype=torch.complex128, device=dev)
    oz, zz = np.ones(1, np.complex128), np.zeros(1, np.complex128)
    ms = time_calls(lambda: L.aoclsparse_zmv(pkg.OP_NONE, pkg._ptr(oz), hz, d0.h, pkg._ptr(xz), pkg._ptr(zz), pkg._ptr(yz)), 30)
    zb = (m + 1 + nnz) * 4 + (2 * m + nnz) * 16
    emit(kind="next", op="aoclsparse_zmv", system="5-pt Laplacian grid %d^2, complex values" % g, ms=round(ms, 4),
         gbs=round(zb / ms / 1e6, 1), frac_of_8TBs=round(zb / ms / 1e6 / 8000, 4))
    L.aoclsparse_destroy(ctypes.byref(hz))
    del A, tcd, tvd
    # symgs, ilu smoother, trsm on the 1000^2 Laplacian
    g = 1000
    m, rp, ci, v = entry.laplace5(g)
    A = pkg.Matrix(0, m, m, rp, ci, v)
    ds = pkg.Descr(mtype=pkg.TYPE_SYMMETRIC, fill=pkg.FILL_LOWER)
    bh, x0 = rng.uniform(-1, 1, m), rng.uniform(-1, 1, m)
    bd, xs = torch.from_numpy(bh).to(dev), torch.from_numpy(x0).to(dev)
    ms = time_calls(lambda: L.aoclsparse_dsymgs(pkg.OP_NONE, A.h, ds.h, 1.0, pkg._ptr(bd), pkg._ptr(xs)), 5, 1)
    o = oracle.dcsr_optimize(m, m, len(v), 0, rp, ci, v)
    t0 = time.time()
    so, xr = oracle.dsymgs(1, 0, 0, 0, 1.0, m, o["val"], o["ind"], o["ptr"], o["idiag"], o["iurow"], bh, x0)
    emit(kind="next", op="aoclsparse_dsymgs (one sweep)", system="5-pt Laplacian grid %d^2" % g, ms=round(ms, 3),
         cpu_serial_ms=round((time.time() - t0) * 1e3, 2))
    pv = ctypes.c_void_p()
    xi = torch.zeros(m, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    t0 = time.time()
    assert L.aoclsparse_dilu_smoother(pkg.OP_NONE, A.h, d0.h, ctypes.byref(pv), None, pkg._ptr(xi), pkg._ptr(bd)) == 0
    torch.cuda.synchronize()
    t_first = time.time() - t0
    ms = time_calls(lambda: L.aoclsparse_dilu_smoother(pkg.OP_NONE, A.h, d0.h, ctypes.byref(pv), None, pkg._ptr(xi), pkg._ptr(bd)), 5, 1)
    t0 = time.time()
    so, lu, dg = oracle.dilu0(m, 0, rp, ci, v)
    t_fac = time.time() - t0
    t0 = time.time()
    so, xr = oracle.dilu_solve(m, 0, dg, lu, rp, ci, bh)
    emit(kind="next", op="aoclsparse_dilu_smoother", system="5-pt Laplacian grid %d^2" % g,
         first_call_s=round(t_first, 3), first_call_includes="GPU factorisation (1,999 levels) + level analysis of both factors",
         apply_ms=round(ms, 3), cpu_factorise_s=round(t_fac, 3), cpu_apply_ms=round((time.time() - t0) * 1e3, 2),
         x_bit_exact=bool(np.array_equal(xi.cpu().numpy(), xr)))
    dl = pkg.Descr(mtype=pkg.TYPE_TRIANGULAR, fill=pkg.FILL_LOWER)
    ms1 = time_calls(lambda: pkg.dtrsv(pkg.OP_NONE, 1.0, A, dl, bd, xi), 5, 1)
    for nr in (8, 64):
        Bd = torch.from_numpy(rng.uniform(-1, 1, (m, nr))).to(dev).contiguous()
        Xd = torch.zeros((m, nr), dtype=torch.float64, device=dev)
        ms = time_calls(lambda: L.aoclsparse_dtrsm(pkg.OP_NONE, 1.0, A.h, dl.h, pkg.ORDER_ROW, pkg._ptr(Bd), nr, nr, pkg._ptr(Xd), nr), 5, 1)
        emit(kind="next", op="aoclsparse_dtrsm, %d right-hand sides (row-major)" % nr, system="lower triangle of the 5-pt Laplacian grid %d^2" % g,
             ms=round(ms, 3), one_trsv_ms=round(ms1, 3),
             note="one launch, columns are the fast grid dimension so their chains advance together; the reference loops trsv per column")
        del Bd, Xd
    del A

if "wider" in what:
    # routines beyond SURVEY 8(f): DIA / BSR products, dense-result sparse product, sparse sum, CSR -> dense, level 1
    import ctypes
    L.aoclsparse_mi355_set_pointer_mode(pkg.PTR_DEVICE)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    one, zero = ctypes.c_double(1.0), ctypes.c_double(0.0)
    g = 4096
    m, rp, ci, v = entry.laplace5(g)
    nnz = len(v)
    rng = np.random.default_rng(5)
    xh = rng.uniform(-1, 1, m)
    so, yref = oracle.dcsrmv(0, 0, 1.0, m, nnz, v, ci, rp, xh, 0.0, np.zeros(m), nthreads=oracle.max_threads())
    # DIA: 5 diagonals, no index array at all
    nd = ctypes.c_int32()
    assert L.aoclsparse_mi355_set_pointer_mode(pkg.PTR_HOST) == 0
    assert L.aoclsparse_csr2dia_ndiag(m, m, d0.h, nnz, pkg._ptr(rp), pkg._ptr(ci), ctypes.byref(nd)) == 0
    off, dv = np.zeros(nd.value, np.int32), np.zeros(nd.value * m)
    t0 = time.perf_counter()
    assert L.aoclsparse_dcsr2dia(m, m, d0.h, pkg._ptr(rp), pkg._ptr(ci), pkg._ptr(v), nd.value, pkg._ptr(off), pkg._ptr(dv)) == 0
    conv_s = time.perf_counter() - t0
    assert L.aoclsparse_mi355_set_pointer_mode(pkg.PTR_DEVICE) == 0
    dvd, offd, xd, yd = t(dv), t(off), t(xh), torch.zeros(m, dtype=torch.float64, device=dev)
    fn = lambda: L.aoclsparse_ddiamv(pkg.OP_NONE, ctypes.byref(one), m, m, nnz, pkg._ptr(dvd), pkg._ptr(offd), nd.value, d0.h,
                                     pkg._ptr(xd), ctypes.byref(zero), pkg._ptr(yd))
    ms = time_calls(fn, 50)
    db = nd.value * m * 8 + nd.value * 4 + 16 * m
    emit(kind="wider", op="aoclsparse_ddiamv (device arrays)", system="5-pt Laplacian grid %d^2, %d diagonals" % (g, nd.value),
         ms=round(ms, 4), dia_bytes=db, gbs=round(db / ms / 1e6, 1), frac_of_8TBs=round(db / ms / 1e6 / 8000, 4),
         gflops=round(2 * nnz / ms / 1e6, 1), csr2dia_host_s=round(conv_s, 3),
         equals_csr_oracle=bool(np.array_equal(yd.cpu().numpy(), yref)))
    # BSR 4x4 on a block-structured matrix: the Laplacian pattern on a (g/2)^2 grid with dense 4x4 blocks
    gb = 1024
    mb, brp, bci, _ = entry.laplace5(gb)
    nblk, dim = len(bci), 4
    bv = rng.uniform(-1, 1, nblk * dim * dim)
    bx = rng.uniform(-1, 1, mb * dim)
    want = oracle.dbsrmv(1.0, mb, dim, 0, bv, bci, brp, bx, 0.0, np.zeros(mb * dim))
    bvd, bcd, bpd, bxd, byd = t(bv), t(bci), t(brp), t(bx), torch.zeros(mb * dim, dtype=torch.float64, device=dev)
    fn = lambda: L.aoclsparse_dbsrmv(pkg.OP_NONE, ctypes.byref(one), mb, mb, dim, pkg._ptr(bvd), pkg._ptr(bcd), pkg._ptr(bpd), d0.h,
                                     pkg._ptr(bxd), ctypes.byref(zero), pkg._ptr(byd))
    ms = time_calls(fn, 50)
    bb = nblk * (dim * dim * 8 + 4) + (mb + 1) * 4 + 16 * mb * dim
    emit(kind="wider", op="aoclsparse_dbsrmv 4x4 (device arrays)", system="5-pt block pattern grid %d^2, %d blocks" % (gb, nblk),
         ms=round(ms, 4), bsr_bytes=bb, gbs=round(bb / ms / 1e6, 1), frac_of_8TBs=round(bb / ms / 1e6 / 8000, 4),
         gflops=round(2 * nblk * dim * dim / ms / 1e6, 1), bit_exact=bool(np.array_equal(byd.cpu().numpy(), want)))
    # dense-result product, sparse sum, CSR -> dense on the 100^2 .. 1000^2 Laplacians
    m2, rp2, ci2, v2 = entry.laplace5(100)
    A2 = pkg.Matrix(0, m2, m2, rp2, ci2, v2)
    Cd = torch.zeros(m2 * m2, dtype=torch.float64, device=dev)
    fn = lambda: L.aoclsparse_dspmmd(pkg.OP_NONE, A2.h, A2.h, pkg.ORDER_ROW, pkg._ptr(Cd), m2)
    ms = time_calls(fn, 20)
    want = oracle.dsp2md((m2, m2, 0, rp2, ci2, v2), False, (m2, m2, 0, rp2, ci2, v2), False, 1.0, 0.0, np.zeros(m2 * m2), True, m2)
    emit(kind="wider", op="aoclsparse_dspmmd (dense C on device)", system="L100 x L100 -> dense 10000^2 (0.8 GB)", ms=round(ms, 4),
         c_fill_gbs=round(m2 * m2 * 8 / ms / 1e6, 1), bit_exact=bool(np.array_equal(Cd.cpu().numpy(), want)))
    Dd = torch.zeros(m2 * m2, dtype=torch.float64, device=dev)
    vd2, rpd2, cid2 = t(v2), t(rp2), t(ci2)
    fn = lambda: L.aoclsparse_dcsr2dense(m2, m2, d0.h, pkg._ptr(vd2), pkg._ptr(rpd2), pkg._ptr(cid2), pkg._ptr(Dd), m2, pkg.ORDER_ROW)
    ms = time_calls(fn, 20)
    emit(kind="wider", op="aoclsparse_dcsr2dense (device arrays)", system="L100 -> dense 10000^2 (0.8 GB)", ms=round(ms, 4),
         fill_gbs=round(m2 * m2 * 8 / ms / 1e6, 1))
    del Cd, Dd
    m3, rp3, ci3, v3 = entry.laplace5(1000)
    A3 = pkg.Matrix(0, m3, m3, rp3, ci3, v3)
    B3 = pkg.Matrix(0, m3, m3, rp3, ci3, rng.uniform(-1, 1, len(v3)))
    L.aoclsparse_mi355_set_pointer_mode(pkg.PTR_AUTO)
    C = ctypes.c_void_p()
    t0 = time.perf_counter()
    assert L.aoclsparse_dadd(pkg.OP_TRANSPOSE, A3.h, 0.5, B3.h, ctypes.byref(C)) == 0
    first = time.perf_counter() - t0
    L.aoclsparse_destroy(ctypes.byref(C))
    t0 = time.perf_counter()
    assert L.aoclsparse_dadd(pkg.OP_TRANSPOSE, A3.h, 0.5, B3.h, ctypes.byref(C)) == 0
    again = time.perf_counter() - t0
    L.aoclsparse_destroy(ctypes.byref(C))
    emit(kind="wider", op="aoclsparse_dadd op=T (host result handle)", system="L1000 (5 M nnz) + same pattern", first_call_s=round(first, 3),
         repeat_call_s=round(again, 3), note="repeat reuses the handles' device copies; the result arrays return to the host")
    # level 1: 16 M entries into a 64 M vector, device arrays
    L.aoclsparse_mi355_set_pointer_mode(pkg.PTR_DEVICE)
    n1, z1 = 1 << 26, 1 << 24
    ix = torch.randperm(n1, device=dev)[:z1].to(torch.int32)
    x1, y1 = torch.rand(z1, dtype=torch.float64, device=dev), torch.rand(n1, dtype=torch.float64, device=dev)
    ms = time_calls(lambda: L.aoclsparse_daxpyi(z1, 0.5, pkg._ptr(x1), pkg._ptr(ix), pkg._ptr(y1)), 20)
    emit(kind="wider", op="aoclsparse_daxpyi (device arrays)", system="16 M random entries of a 64 M vector", ms=round(ms, 4),
         gbs_algorithmic=round(z1 * 28 / ms / 1e6, 1), gbs_sectors=round(z1 * (12 + 64) / ms / 1e6, 1))
    ms = time_calls(lambda: L.aoclsparse_dgthr(z1, pkg._ptr(y1), pkg._ptr(x1), pkg._ptr(ix)), 20)
    emit(kind="wider", op="aoclsparse_dgthr (device arrays)", system="16 M random entries of a 64 M vector", ms=round(ms, 4),
         gbs_algorithmic=round(z1 * 20 / ms / 1e6, 1), gbs_sectors=round(z1 * (12 + 32) / ms / 1e6, 1))
    ms = time_calls(lambda: L.aoclsparse_ddoti(z1, pkg._ptr(x1), pkg._ptr(ix), pkg._ptr(y1)), 20)
    emit(kind="wider", op="aoclsparse_ddoti (device arrays, result by value)", system="16 M random entries of a 64 M vector",
         ms=round(ms, 4), gbs_algorithmic=round(z1 * 20 / ms / 1e6, 1))
    L.aoclsparse_mi355_set_pointer_mode(pkg.PTR_AUTO)

if "setup" in what:
    # one-off costs on the path: optimize with an mv hint (clean-CSR checks on the host,
    # upload, SELL-64 build on the GPU) and sp2m (A*A), each beside the restated CPU routine on this host
    import ctypes
    L.aoclsparse_mi355_set_pointer_mode(pkg.PTR_AUTO)
    for g in (1000, 4096):
        m, rp, ci, v = entry.laplace5(g)
        nnz = len(v)
        t = time.time()
        A = pkg.Matrix(0, m, m, rp, ci, v)
        t_create = time.time() - t
        assert L.aoclsparse_set_mv_hint(A.h, pkg.OP_NONE, d0.h, 100) == 0
        t = time.time()
        assert L.aoclsparse_optimize(A.h) == 0
        torch.cuda.synchronize()
        t_opt = time.time() - t
        x = np.ones(m)
        y = np.zeros(m)
        t = time.time()
        assert pkg.dmv(pkg.OP_NONE, 1.0, A, d0, x, 0.0, y) == 0
        t_first = time.time() - t
        t = time.time()
        o = oracle.dcsr_optimize(m, m, nnz, 0, rp, ci, v)
        t_cpu = time.time() - t
        emit(kind="setup", op="create + set_mv_hint + optimize", system="5-pt Laplacian grid %d^2 (nnz=%d)" % (g, nnz),
             create_s=round(t_create, 4), optimize_s=round(t_opt, 4), first_mv_host_arrays_s=round(t_first, 4),
             cpu_clean_csr_restatement_s=round(t_cpu, 4), spmv_calls_to_amortise=int(t_opt / 0.22e-3) if g == 4096 else None)
        del A
    for g in (300, 1000):
        m, rp, ci, v = entry.laplace5(g)
        A = pkg.Matrix(0, m, m, rp, ci, v)
        C = ctypes.c_void_p()
        assert L.aoclsparse_sp2m(pkg.OP_NONE, d0.h, A.h, pkg.OP_NONE, d0.h, A.h, pkg.STAGE_FULL, ctypes.byref(C)) == 0  # warm-up (uploads A)
        L.aoclsparse_destroy(ctypes.byref(C))
        best = 1e9
        for _ in range(3):
            t = time.time()
            assert L.aoclsparse_sp2m(pkg.OP_NONE, d0.h, A.h, pkg.OP_NONE, d0.h, A.h, pkg.STAGE_FULL, ctypes.byref(C)) == 0
            best = min(best, time.time() - t)
            nb, cm, cn, cz = ctypes.c_int(), ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
            pr, pc_, pv = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
            assert L.aoclsparse_export_dcsr(C, ctypes.byref(nb), ctypes.byref(cm), ctypes.byref(cn), ctypes.byref(cz),
                                            ctypes.byref(pr), ctypes.byref(pc_), ctypes.byref(pv)) == 0
            L.aoclsparse_destroy(ctypes.byref(C))
        t = time.time()
        so, pcr, icr, vcr = oracle.dcsr2m(m, m, 0, rp, ci, v, 0, rp, ci, v)
        t_cpu = time.time() - t
        emit(kind="setup", op="aoclsparse_sp2m A*A (result returned as host CSR)", system="5-pt Laplacian grid %d^2" % g,
             nnz_c=int(cz.value), s=round(best, 4), cpu_restatement_s=round(t_cpu, 4), same_nnz=bool(cz.value == len(icr)))
        del A

if "pcie" in what:
    L.aoclsparse_mi355_set_pointer_mode(pkg.PTR_AUTO)
    g = 4096
    m, rp, ci, v = entry.laplace5(g)
    nnz = len(v)
    A = pkg.Matrix(0, m, m, rp, ci, v)
    assert L.aoclsparse_set_mv_hint(A.h, pkg.OP_NONE, d0.h, 10) == 0 and L.aoclsparse_optimize(A.h) == 0
    xh = np.sin(0.01 * np.arange(m))
    yh = np.zeros(m)
    pkg.dmv(pkg.OP_NONE, 1.0, A, d0, xh, 0.0, yh)
    t = time.perf_counter()
    reps = 5
    for _ in range(reps):
        pkg.dmv(pkg.OP_NONE, 1.0, A, d0, xh, 0.0, yh)
    dt = (time.perf_counter() - t) / reps
    emit(kind="pcie-inclusive", workload="aoclsparse_dmv, HOST x and y (pageable), grid 4096^2 Laplacian resident in HBM",
         ms=round(dt * 1e3, 3), gflops=round(2.0 * nnz / dt / 1e9, 2),
         note="x H2D + y D2H of 134 MB each per call dominate; kernel itself ~0.25 ms")
    # one-shot raw API with everything on the host: the matrix crosses PCIe every call
    yh2 = np.zeros(m)
    pkg.dcsrmv(pkg.OP_NONE, 1.0, m, m, nnz, v, ci, rp, d0, xh, 0.0, yh2)
    t = time.perf_counter()
    pkg.dcsrmv(pkg.OP_NONE, 1.0, m, m, nnz, v, ci, rp, d0, xh, 0.0, yh2)
    dt = time.perf_counter() - t
    emit(kind="pcie-inclusive", workload="aoclsparse_dcsrmv, ALL arrays on the host (matrix re-sent every call)",
         ms=round(dt * 1e3, 3), gflops=round(2.0 * nnz / dt / 1e9, 2), equal_to_handle_path=bool(np.array_equal(yh, yh2)))
