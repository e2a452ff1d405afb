"""GPU tier, round 4: argument checks of the column-sharded csrmm entry points on the WHOLE operands, the process-wide beta = 0
mode word, the in-library multi-device bookkeeping (same-device honesty, per-device times), and the kernels added this round.
Every comparison is against the CPU oracle through the C ABI; tolerances are written where they are used."""
import json
import os
import subprocess
import sys
import textwrap

import ctypes
import numpy as np
import pytest

import oracle
from util import EPS64, ROOT, beta0_overwrite, laplace5, pkg, random_csr

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
P = pkg()
L = P.lib()
HERE = os.path.dirname(os.path.abspath(__file__))
INVALID_SIZE = [k for k, v in P.STATUS.items() if v.endswith("invalid_size")][0]


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_sharded_entry_points_validate_the_whole_operands():
    """A row-major call with ldb or ldc < n returns invalid_size in the reference (csrmm.hpp:592-611) whatever thread computes
    which columns.  The column-sharded entry points must return that status too -- a shard's own width (n / world) would let it
    through and compute on overlapping rows -- and must do so before any replica is built."""
    m, k, n = 500, 400, 64
    rp, ci, v = random_csr(11, m, k, lambda r, i: r.integers(0, 7))
    A = P.Matrix(0, m, k, rp, ci, v)
    d = P.Descr()
    rng = np.random.default_rng(2)
    B, C = rng.uniform(-1, 1, k * n), np.zeros(m * n)
    st, dev0, _, _ = P.device_info()
    assert st == 0
    for ldb, ldc in ((n // 2, n), (n, n // 2), (n - 1, n - 1)):
        want = P.dcsrmm(P.OP_NONE, 1.0, A, d, P.ORDER_ROW, B, n, ldb, 0.0, C, ldc)
        assert want == INVALID_SIZE
        for world in (2, 4):
            for rank in range(world):
                assert P.dcsrmm_shard(P.OP_NONE, 1.0, A, d, P.ORDER_ROW, B, n, ldb, 0.0, C, ldc, world, rank) == want
        assert P.dcsrmm_multi(P.OP_NONE, 1.0, A, d, P.ORDER_ROW, B, n, ldb, 0.0, C, ldc, [dev0] * 4) == want
    assert L.aoclsparse_mi355_replica_count(A.h) == 0  # rejected before any replica existed
    # column-major: ld against the ROW counts, LP64 range of n * ld
    assert P.dcsrmm_shard(P.OP_NONE, 1.0, A, d, P.ORDER_COLUMN, B, n, k - 1, 0.0, C, m, 2, 1) == INVALID_SIZE
    assert P.dcsrmm_multi(P.OP_NONE, 1.0, A, d, P.ORDER_COLUMN, B, n, k, 0.0, C, m - 1, [dev0] * 2) == INVALID_SIZE
    # and the valid call still equals the single-call product bit for bit
    ref = C.copy()
    assert P.dcsrmm(P.OP_NONE, 1.5, A, d, P.ORDER_ROW, B, n, n, 0.0, ref, n) == 0
    got = C.copy()
    assert P.dcsrmm_multi(P.OP_NONE, 1.5, A, d, P.ORDER_ROW, B, n, n, 0.0, got, n, [dev0] * 3) == 0
    assert np.array_equal(got, ref)
    # a csrmm replica carries the mm hint only: a handle with sv + mv hints does not build TRSV / SELL plans on the other slots
    cnt = L.aoclsparse_mi355_multi_last_ms(None, 0)
    assert cnt == 3
    import ctypes
    buf = (ctypes.c_float * 8)()
    assert L.aoclsparse_mi355_multi_last_ms(buf, 8) == 3 and all(buf[i] > 0 for i in range(3))


def test_beta0_mode_setter_is_not_overridden_by_the_environment():
    """AOCLSPARSE_MI355_CSRMM_BETA0_OVERWRITE=1 seeds the mode once; an explicit set(0) made BEFORE the first product must win
    (the lazy read used to override it).  Observable: with C = NaN and beta = 0 the default mode propagates NaN
    (the reference's arithmetic, csrmm.hpp:83,129), the overwrite mode does not."""
    code = textwrap.dedent("""
        import sys, numpy as np
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        from util import pkg, random_csr
        P = pkg(); L = P.lib()
        m, k, n = 300, 300, 8
        rp, ci, v = random_csr(5, m, k, lambda r, i: 1 + r.integers(0, 5))
        A = P.Matrix(0, m, k, rp, ci, v); d = P.Descr()
        B = np.ones(k * n)
        mode = sys.argv[1]
        if mode != "env":
            assert L.aoclsparse_mi355_set_csrmm_beta0_overwrite(int(mode)) == 0
        C = np.full(m * n, np.nan)
        assert P.dcsrmm(P.OP_NONE, 1.0, A, d, P.ORDER_ROW, B, n, n, 0.0, C, n) == 0
        print("nan" if np.isnan(C).any() else "clean")
    """) % (HERE, ROOT)
    for envv, mode, want in (("1", "env", "clean"), ("1", "0", "nan"), ("0", "1", "clean"), (None, "env", "nan")):
        env = {k: v for k, v in os.environ.items() if k != "AOCLSPARSE_MI355_CSRMM_BETA0_OVERWRITE"}
        if envv is not None:
            env["AOCLSPARSE_MI355_CSRMM_BETA0_OVERWRITE"] = envv
        r = subprocess.run([sys.executable, "-c", code, mode], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and r.stdout.strip().endswith(want), (envv, mode, r.stdout, r.stderr[-1500:])


def test_multi_check_reports_no_efficiency_on_one_device():
    """tools/multi_check.py with every slot on device 0 (all a one-GPU box can do): control flow and bit-identity only --
    `same_device: true` and NO efficiency figure (two slots sharing one GPU say nothing about scaling).  The slabs
    are filled on the null stream right before the call without a synchronize: the slot streams are blocking streams."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "multi_check.py"), "--devices", "2", "--same-device", "--grid", "300",
                        "--cols", "64", "--reps", "3"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert res["same_device"] is True and "efficiency_wall" not in res and res["slabs_bit_exact"] is True
    assert res["devices"] == [res["devices"][0]] * 2 and res["replicas"] == 1


# --------------------------------------------------------------------------------------------------
# column-major csrmm, LDS-window kernel (csrmm_window_kernels.hip)
# --------------------------------------------------------------------------------------------------
def _banded(g, rows, pattern, seed, base=0, extra_long=0, empty_every=0):
    """m = g * rows rows; row r holds the offsets of `pattern` (relative columns, clipped to [0, m)) in increasing order,
    values U(-1, 1).  extra_long: that many rows additionally get 14 more entries inside the band (rows longer than the
    kernel's register cache); empty_every: every such row is left empty."""
    rng = np.random.default_rng(seed)
    m = g * rows
    longs = set(int(t) for t in rng.choice(m, size=extra_long, replace=False)) if extra_long else set()
    rp = np.zeros(m + 1, np.int64)
    cols = []
    pat = np.array(sorted(pattern), np.int64)
    for r in range(m):
        c = r + pat
        c = c[(c >= 0) & (c < m)]
        if r in longs:
            c = np.unique(np.concatenate([c, np.clip(r + rng.integers(-g, g, size=14), 0, m - 1)]))
        if empty_every and r % empty_every == empty_every - 1:
            c = c[:0]
        cols.append(c)
        rp[r + 1] = rp[r] + len(c)
    ci = np.concatenate(cols)
    v = rng.uniform(-1, 1, len(ci))
    return m, (rp + base).astype(np.int32), (ci + base).astype(np.int32), v


def _same_bits(a, b):
    """bit-for-bit equality (signs of zeros included) where the values are numbers; NaN must meet NaN (its payload / sign are the
    host FPU's or the GPU's default and are not part of the reference's arithmetic)"""
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    na, nb = np.isnan(a), np.isnan(b)
    u = np.uint64 if a.dtype == np.float64 else np.uint32
    return bool(np.array_equal(na, nb) and np.array_equal(a.view(u)[~na], b.view(u)[~nb]))


@pytest.mark.parametrize("case", ["laplace5", "nine_point_base1", "long_and_empty_rows"])
def test_csrmm_column_major_window_kernel_bit_exact(case):
    """csrmm_colwin_kernel (column-major operands, banded A: each B column's stretch staged in LDS by LDS-DMA, the rows' entries
    in registers) against oracle.dcsrmm (csrmm_col_major_ref, csrmm.hpp:36-90): bit for bit, for 5-point rows (8 rows per lane of 512),
    9-point rows in base 1 (4 rows per lane), rows longer than the register cache and empty rows; column counts that are not a
    multiple of the 64-column chunk, odd counts, padded leading dimensions, alpha / beta classes, both beta = 0 modes, Inf / NaN
    in B next to short rows (padding entries must not pick them up), m not a multiple of the rows per workgroup."""
    if case == "laplace5":
        g = 150
        m, rp, ci, v = laplace5(g)
        base, want_rows = 0, 4096
    elif case == "nine_point_base1":
        g = 101
        m, rp, ci, v = _banded(g, 97, [-g - 1, -g, -g + 1, -1, 0, 1, g - 1, g, g + 1], 5, base=1)
        base, want_rows = 1, 2048
    else:
        g = 120
        m, rp, ci, v = _banded(g, 90, [-g, -1, 0, 1, g], 6, extra_long=40, empty_every=37)
        base, want_rows = 0, 2048
    A = P.Matrix(base, m, m, rp, ci, v)
    d = P.Descr(base=base)
    rng = np.random.default_rng(12)
    first = True
    for n, ldb, ldc, alpha, beta in ((70, m, m, 1.0, 0.0), (5, m + 2, m + 6, -1.5, 0.75), (64, m, m + 1, 2.0, 0.0), (33, m + 4, m, 1.0, -1.0)):
        B = rng.uniform(-1, 1, ldb * n)
        # Inf / NaN in B: rows that reference them must produce the reference's Inf / NaN, every other row must not see them
        Bm = B.reshape(n, ldb)
        Bm[0, 0], Bm[1, m // 2], Bm[n - 1, m - 1] = np.inf, np.nan, -np.inf
        C0 = rng.uniform(-1, 1, ldc * n)
        so, Cr = oracle.dcsrmm("col", alpha, base, v, ci, rp, m, B, n, ldb, beta, C0, ldc)
        assert so == 0
        for overwrite in ((False, True) if beta == 0.0 else (False,)):
            Cd = dev(C0)
            assert L.aoclsparse_mi355_set_csrmm_beta0_overwrite(1 if overwrite else 0) == 0
            try:
                assert P.dcsrmm(P.OP_NONE, alpha, A, d, P.ORDER_COLUMN, dev(B), n, ldb, beta, Cd, ldc) == 0
                torch.cuda.synchronize()
            finally:
                assert L.aoclsparse_mi355_set_csrmm_beta0_overwrite(0) == 0
            assert _same_bits(Cd.cpu().numpy(), Cr), (case, n, ldb, ldc, alpha, beta, overwrite)
        if first:
            assert A.spmv_info().mm_window_rows == want_rows, "the window kernel was not selected"
            first = False
    # operands the window kernel cannot take (odd leading dimension: a column's stretch would not start on 16 bytes) fall back to
    # the lane-per-row kernels with the same bits
    n, ldb = 12, m + 1
    B = rng.uniform(-1, 1, ldb * n)
    C0 = rng.uniform(-1, 1, m * n)
    so, Cr = oracle.dcsrmm("col", 1.0, base, v, ci, rp, m, B, n, ldb, 0.5, C0, m)
    Cd = dev(C0)
    assert P.dcsrmm(P.OP_NONE, 1.0, A, d, P.ORDER_COLUMN, dev(B), n, ldb, 0.5, Cd, m) == 0
    torch.cuda.synchronize()
    assert np.array_equal(Cd.cpu().numpy(), Cr)


def test_csrmm_column_major_window_kernel_float_and_nan_in_c():
    """float operands (4 elements per 16-byte piece) against the serial fp32 chain of csrmm_col_major_ref on sampled rows; NaN
    already in C: propagated by default as in the reference (csrmm.hpp:83), overwritten in the opt-in mode."""
    g = 160
    m, rp, ci, v = laplace5(g)
    vf = (v + np.random.default_rng(3).uniform(-0.1, 0.1, len(v))).astype(np.float32)
    Af = P.Matrix(0, m, m, rp, ci, vf)
    d = P.Descr()
    n = 24
    rng = np.random.default_rng(4)
    Bf = rng.uniform(-1, 1, m * n).astype(np.float32)
    Cf = torch.zeros(m * n, dtype=torch.float32, device="cuda")
    assert L.aoclsparse_scsrmm(P.OP_NONE, 1.0, Af.h, d.h, P.ORDER_COLUMN, P._ptr(dev(Bf)), n, m, 0.0, P._ptr(Cf), m) == 0
    torch.cuda.synchronize()
    assert Af.spmv_info().mm_window_rows == 4096
    got = Cf.cpu().numpy().reshape(n, m)
    Bm = Bf.reshape(n, m)
    for i in list(range(0, m, 211)) + [m - 1]:
        acc = np.zeros(n, np.float32)
        for p in range(rp[i], rp[i + 1]):  # fmaf chain: exact product + sum in double, one rounding to float
            acc = (np.float64(vf[p]) * Bm[:, ci[p]].astype(np.float64) + acc.astype(np.float64)).astype(np.float32)
        assert np.array_equal(got[:, i], acc), i
    Ad = P.Matrix(0, m, m, rp, ci, v)
    B = rng.uniform(-1, 1, m * n)
    Cd = torch.full((m * n,), float("nan"), dtype=torch.float64, device="cuda")
    assert P.dcsrmm(P.OP_NONE, 1.0, Ad, d, P.ORDER_COLUMN, dev(B), n, m, 0.0, Cd, m) == 0
    torch.cuda.synchronize()
    assert bool(torch.isnan(Cd).all())
    with beta0_overwrite(P):
        assert P.dcsrmm(P.OP_NONE, 1.0, Ad, d, P.ORDER_COLUMN, dev(B), n, m, 0.0, Cd, m) == 0
        torch.cuda.synchronize()
    so, Cr = oracle.dcsrmm("col", 1.0, 0, v, ci, rp, m, B, n, m, 0.0, np.zeros(m * n), m)
    assert np.array_equal(Cd.cpu().numpy(), Cr)


# --------------------------------------------------------------------------------------------------
# blocked-ELL + MFMA csrmm (csrmm_bell_kernels.hip): block-dense matrices
# --------------------------------------------------------------------------------------------------
sys.path.insert(0, os.path.join(ROOT, "tools"))
import standins  # noqa: E402


def _submatrix(m, rp, ci, v, mm, kk):
    """leading mm x kk part of a CSR matrix"""
    rows = np.repeat(np.arange(m), np.diff(rp))
    sel = (rows < mm) & (ci < kk)
    rp2 = np.zeros(mm + 1, np.int64)
    np.add.at(rp2, rows[sel] + 1, 1)
    return np.cumsum(rp2).astype(np.int32), ci[sel].copy(), v[sel].copy()


@pytest.mark.parametrize("keep", [1.0, 0.75])
def test_csrmm_blocked_ell_mfma_bit_exact(keep):
    """A block-dense matrix (16 unknowns per node, 7-point node stencil: 16 x 16 tiles full, or thinned to 75 %) gets a blocked-ELL
    copy from aoclsparse_optimize (mm hint) and runs on v_mfma_f64_16x16x4_f64.  The instruction accumulates k upwards as one FMA
    chain per element, so the result must equal oracle.dcsrmm (csrmm.hpp:36-90) BIT FOR BIT: row-major with column counts that are
    and are not multiples of the 16-column tile, padded leading dimensions, alpha / beta classes and both beta = 0 modes, a row
    count and a column count that are not multiples of 16, and column-major operands (the transposed product: D' = B^T A^T, C stored in column
    segments)."""
    m0, rp0, ci0, v0 = standins.block_dense(6, 5, 4, keep=keep, seed=9)
    for mm, kk in ((m0, m0), (m0 - 5, m0 - 9)):
        rp, ci, v = (rp0, ci0, v0) if mm == m0 else _submatrix(m0, rp0, ci0, v0, mm, kk)
        A = P.Matrix(0, mm, kk, rp, ci, v)
        d = P.Descr()
        assert L.aoclsparse_set_mm_hint(A.h, P.OP_NONE, d.h, 10) == 0 and L.aoclsparse_optimize(A.h) == 0
        info = A.spmv_info()
        assert info.mm_bell_width == 7, "the blocked-ELL copy was not built"
        assert abs(info.mm_bell_fill_permille - 1000 * len(v) / (256.0 * _blocks(mm, rp, ci))) <= 1
        rng = np.random.default_rng(21)
        for n, ldb, ldc, alpha, beta in ((64, 64, 64, 1.0, 0.0), (40, 44, 41, -0.5, 1.25), (16, 16, 16, 2.0, 0.0), (72, 72, 80, 1.0, -1.0),
                                         (7, 7, 7, 1.0, 0.0)):
            B = rng.uniform(-1, 1, kk * ldb)
            C0 = rng.uniform(-1, 1, mm * ldc)
            Bc = np.ascontiguousarray(B.reshape(kk, ldb)[:, :n].T).ravel()
            Cc = np.ascontiguousarray(C0.reshape(mm, ldc)[:, :n].T).ravel()
            so, Cr = oracle.dcsrmm("col", alpha, 0, v, ci, rp, mm, Bc, n, kk, beta, Cc, mm)
            assert so == 0
            ref = Cr.reshape(n, mm).T
            for overwrite in ((False, True) if beta == 0.0 else (False,)):
                Cd = dev(C0)
                assert L.aoclsparse_mi355_set_csrmm_beta0_overwrite(1 if overwrite else 0) == 0
                try:
                    assert P.dcsrmm(P.OP_NONE, alpha, A, d, P.ORDER_ROW, dev(B), n, ldb, beta, Cd, ldc) == 0
                    torch.cuda.synchronize()
                finally:
                    assert L.aoclsparse_mi355_set_csrmm_beta0_overwrite(0) == 0
                got = Cd.cpu().numpy().reshape(mm, ldc)
                assert _same_bits(got[:, :n], ref), (keep, mm, n, ldb, ldc, alpha, beta, overwrite)
                assert np.array_equal(got[:, n:], C0.reshape(mm, ldc)[:, n:])  # padding untouched
        # column-major operands: the transposed MFMA product (csrmm_bell_mfma_col_kernel), C stored in column segments
        for n, ldb, ldc, alpha, beta in ((64, kk, mm, 1.0, 0.0), (48, kk + 3, mm + 5, 1.5, 0.5), (21, kk, mm, -1.0, 0.0), (7, kk + 1, mm, 1.0, 2.0)):
            B = rng.uniform(-1, 1, ldb * n)
            C0 = rng.uniform(-1, 1, ldc * n)
            so, Cr = oracle.dcsrmm("col", alpha, 0, v, ci, rp, mm, B, n, ldb, beta, C0, ldc)
            assert so == 0
            for overwrite in ((False, True) if beta == 0.0 else (False,)):
                Cd = dev(C0)
                assert L.aoclsparse_mi355_set_csrmm_beta0_overwrite(1 if overwrite else 0) == 0
                try:
                    assert P.dcsrmm(P.OP_NONE, alpha, A, d, P.ORDER_COLUMN, dev(B), n, ldb, beta, Cd, ldc) == 0
                    torch.cuda.synchronize()
                finally:
                    assert L.aoclsparse_mi355_set_csrmm_beta0_overwrite(0) == 0
                assert _same_bits(Cd.cpu().numpy(), Cr), (keep, mm, "col", n, ldb, ldc, alpha, beta, overwrite)


def test_csrmm_blocked_ell_inf_nan_in_b_and_c_as_the_reference():
    """The tile of the MFMA kernel multiplies its explicit zeros into the sum: an Inf / NaN in B at a column a row does NOT store
    would give 0 * Inf = NaN where the reference's CSR kernel (csrmm.hpp:69-85) never looks.  Round 5: an element whose tile sum is
    not finite is recomputed from the CSR arrays, so the product is the reference's for every B -- finite where the reference is
    finite, Inf / NaN exactly where the row's own entries meet them.  With 75 % fill every row has padded positions.  Also: an
    Inf / NaN already in C propagates through beta = 0 (the reference's 0 * C) in the default mode and is overwritten in the opt-in
    mode; both layouts."""
    mm, rp, ci, v = standins.block_dense(6, 5, 4, keep=0.75, seed=12)
    A = P.Matrix(0, mm, mm, rp, ci, v)
    d = P.Descr()
    assert L.aoclsparse_set_mm_hint(A.h, P.OP_NONE, d.h, 10) == 0 and L.aoclsparse_optimize(A.h) == 0
    assert A.spmv_info().mm_bell_width > 0
    rng = np.random.default_rng(77)
    n = 48
    for layout in ("row", "col"):
        B = rng.uniform(-1, 1, (mm, n))
        bad_rows = rng.choice(mm, size=40, replace=False)
        B[bad_rows[:20], rng.integers(0, n, 20)] = np.inf
        B[bad_rows[20:30], rng.integers(0, n, 10)] = -np.inf
        B[bad_rows[30:], rng.integers(0, n, 10)] = np.nan
        C0 = rng.uniform(-1, 1, (mm, n))
        C0[5, 3], C0[77, 40], C0[300, 0] = np.inf, np.nan, -np.inf
        Bc, Cc = np.ascontiguousarray(B.T).ravel(), np.ascontiguousarray(C0.T).ravel()
        for alpha, beta, overwrite in ((1.0, 0.0, False), (1.0, 0.0, True), (-2.0, 0.5, False)):
            Cin = Cc.copy()
            if overwrite:  # BLAS convention: whatever C held is gone -- the reference arithmetic on a finite C
                Cin = np.where(np.isfinite(Cin), Cin, 0.25)
            so, Cr = oracle.dcsrmm("col", alpha, 0, v, ci, rp, mm, Bc, n, mm, beta, Cin, mm)
            assert so == 0
            ref = Cr.reshape(n, mm).T
            assert np.isfinite(ref).sum() > 0.5 * ref.size and (~np.isfinite(ref)).sum() > 20
            assert L.aoclsparse_mi355_set_csrmm_beta0_overwrite(1 if overwrite else 0) == 0
            try:
                if layout == "row":
                    Cd = dev(np.ascontiguousarray(C0).ravel())
                    assert P.dcsrmm(P.OP_NONE, alpha, A, d, P.ORDER_ROW, dev(np.ascontiguousarray(B).ravel()), n, n, beta, Cd, n) == 0
                    torch.cuda.synchronize()
                    got = Cd.cpu().numpy().reshape(mm, n)
                else:
                    Cd = dev(Cc)
                    assert P.dcsrmm(P.OP_NONE, alpha, A, d, P.ORDER_COLUMN, dev(Bc), n, mm, beta, Cd, mm) == 0
                    torch.cuda.synchronize()
                    got = Cd.cpu().numpy().reshape(n, mm).T
            finally:
                assert L.aoclsparse_mi355_set_csrmm_beta0_overwrite(0) == 0
            fin = np.isfinite(ref)
            assert np.array_equal(np.isfinite(got), fin), (layout, alpha, beta, overwrite, int((np.isfinite(got) != fin).sum()))
            assert np.array_equal(got[fin], ref[fin])  # bit for bit where the reference is finite
            assert np.array_equal(np.isnan(got), np.isnan(ref)) and np.array_equal(got[np.isinf(ref)], ref[np.isinf(ref)])


def _blocks(m, rp, ci):
    rows = np.repeat(np.arange(m), np.diff(rp))
    return len(np.unique((rows // 16).astype(np.int64) * (1 << 32) + ci // 16))


def test_csrmm_blocked_ell_is_not_chosen_below_half_fill_or_for_unsorted_rows():
    """The format choice follows the reference's kind of rule (a fill threshold, convert.cpp:36-147): tiles less than half full, or
    rows that are not sorted (the tile walks k upwards: only sorted rows give the CSR-order chain) keep the CSR kernels -- with
    the same bits."""
    m, rp, ci, v = standins.block_dense(6, 5, 4, keep=0.35, seed=10)
    A = P.Matrix(0, m, m, rp, ci, v)
    d = P.Descr()
    assert L.aoclsparse_set_mm_hint(A.h, P.OP_NONE, d.h, 10) == 0 and L.aoclsparse_optimize(A.h) == 0
    assert A.spmv_info().mm_bell_width == 0
    m, rp, ci, v = standins.block_dense(6, 5, 4, seed=11)
    ci2 = ci.copy()
    s, e = rp[100], rp[101]
    ci2[s:e] = ci[s:e][::-1]  # one row in descending order
    v2 = v.copy()
    A2 = P.Matrix(0, m, m, rp, ci2, v2)
    assert L.aoclsparse_set_mm_hint(A2.h, P.OP_NONE, d.h, 10) == 0 and L.aoclsparse_optimize(A2.h) == 0
    n = 32
    rng = np.random.default_rng(2)
    B, C0 = rng.uniform(-1, 1, m * n), rng.uniform(-1, 1, m * n)
    Cd = dev(C0)
    assert P.dcsrmm(P.OP_NONE, 1.0, A2, d, P.ORDER_ROW, dev(B), n, n, 0.0, Cd, n) == 0
    torch.cuda.synchronize()
    assert A2.spmv_info().mm_bell_width == 0
    Bc = np.ascontiguousarray(B.reshape(m, n).T).ravel()
    so, Cr = oracle.dcsrmm("col", 1.0, 0, v2, ci2, rp, m, Bc, n, m, 0.0, np.zeros(m * n), m)
    assert np.array_equal(Cd.cpu().numpy().reshape(m, n), Cr.reshape(n, m).T)


# --------------------------------------------------------------------------------------------------
# the analysed device state of a handle travels; the library's own RCCL communicator
# --------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("which", ["laplace", "block_dense", "random"])
def test_mm_state_export_adopt_round_trip(which):
    """aoclsparse_mi355_mm_state_export -> device copies of the buffers -> aoclsparse_mi355_mm_state_adopt: the new handle has
    done no analysis, yet every csrmm (both layouts, wide and narrow) gives the exporter's bits, its plans are the exporter's
    (window kernel / blocked-ELL copy / row groups), and the rest of the API works on it (export, ?mv, a second export)."""
    if which == "laplace":
        m, rp, ci, v = laplace5(140)
    elif which == "block_dense":
        m, rp, ci, v = standins.block_dense(5, 4, 4, seed=3)
    else:
        m = 5000
        rp, ci, v = random_csr(77, m, m, lambda r, i: r.integers(0, 14))
    A = P.Matrix(0, m, m, rp, ci, v)
    d = P.Descr()
    assert L.aoclsparse_set_mm_hint(A.h, P.OP_NONE, d.h, 10) == 0 and L.aoclsparse_optimize(A.h) == 0
    st, state, ptrs = A.mm_state_export()
    assert st == 0 and list(state.bytes)[:3] == [4 * (m + 1), 4 * len(v), 8 * len(v)]
    # what a receiving rank holds: its OWN device copies of the buffers (here: clones made through torch)
    from aocl_sparse_amd.sharded import _DeviceView
    held = [torch.as_tensor(_DeviceView(p, n), device="cuda").clone() if n else None for p, n in zip(ptrs, list(state.bytes))]
    torch.cuda.synchronize()
    st, R = P.Matrix.mm_state_adopt(state, [t.data_ptr() if t is not None else None for t in held])
    assert st == 0 and (R.m, R.n, R.nnz, R.base) == (m, m, len(v), 0)
    del held
    ia, ir = A.spmv_info(), R.spmv_info()
    for f in ("row_blocks", "tile", "max_row_nnz", "mm_groups", "mm_window_rows", "mm_bell_width", "mm_bell_fill_permille"):
        assert getattr(ia, f) == getattr(ir, f), f
    if which == "laplace":
        assert ir.mm_window_rows == 4096
    if which == "block_dense":
        assert ir.mm_bell_width == 7
    e = R.export()
    assert e["status"] == 0 and np.array_equal(e["row_ptr"], rp) and np.array_equal(e["col_ind"], ci) and np.array_equal(e["val"], v)
    rng = np.random.default_rng(5)
    for order, n in ((P.ORDER_COLUMN, 24), (P.ORDER_ROW, 160), (P.ORDER_ROW, 32), (P.ORDER_COLUMN, 6)):
        ld = m if order == P.ORDER_COLUMN else n
        B, C0 = rng.uniform(-1, 1, m * n), rng.uniform(-1, 1, m * n)
        Ca, Cr = dev(C0), dev(C0)
        assert P.dcsrmm(P.OP_NONE, 1.25, A, d, order, dev(B), n, ld, -0.5, Ca, ld) == 0
        assert P.dcsrmm(P.OP_NONE, 1.25, R, d, order, dev(B), n, ld, -0.5, Cr, ld) == 0
        torch.cuda.synchronize()
        assert torch.equal(Ca, Cr), (which, order, n)
    x = rng.uniform(-1, 1, m)
    ya, yr = np.zeros(m), np.zeros(m)
    assert P.dmv(P.OP_NONE, 1.0, A, d, x, 0.0, ya) == 0 and P.dmv(P.OP_NONE, 1.0, R, d, x, 0.0, yr) == 0
    assert np.array_equal(ya, yr)
    st2, state2, _ = R.mm_state_export()
    assert st2 == 0 and list(state2.bytes) == list(state.bytes) and list(state2.scalars) == list(state.scalars)
    # a corrupted header is refused
    bad = P.MmState()
    ctypes_copy = bytes(state)
    import ctypes
    ctypes.memmove(ctypes.addressof(bad), ctypes_copy, len(ctypes_copy))
    bad.scalars[0] = 1
    st, none = P.Matrix.mm_state_adopt(bad, [None] * P.MM_STATE_BUFFERS)
    assert st != 0 and none is None
    # ... and so is a state whose buffers are smaller than the plans its scalars announce (a truncated / mismatched transfer):
    # every announced plan, one mutation each -- the receiver refuses BEFORE any kernel could index past a buffer
    held = [torch.as_tensor(_DeviceView(p, n), device="cuda").clone() if n else None for p, n in zip(ptrs, list(state.bytes))]
    addrs = [t.data_ptr() if t is not None else None for t in held]

    def mutated(fn):
        b = P.MmState()
        ctypes.memmove(ctypes.addressof(b), ctypes_copy, len(ctypes_copy))
        fn(b)
        return P.Matrix.mm_state_adopt(b, addrs)

    S_NBLOCKS, S_TILE, S_WIN, S_WIN_ROWS, S_BELL, S_BELL_WIDTH, S_GROUPS_VALID, S_NGROUPS = 8, 11, 21, 22, 23, 25, 17, 15
    cases = [("row blocks", lambda b: b.bytes.__setitem__(3, b.bytes[3] - 8)),
             ("block count", lambda b: b.scalars.__setitem__(S_NBLOCKS, b.scalars[S_NBLOCKS] + 1)),
             ("tile", lambda b: b.scalars.__setitem__(S_TILE, 768)),
             ("CSR values", lambda b: b.bytes.__setitem__(2, b.bytes[2] - 8))]
    if state.scalars[S_WIN]:
        cases += [("window table", lambda b: b.bytes.__setitem__(9, b.bytes[9] - 8)),
                  ("window rows", lambda b: b.scalars.__setitem__(S_WIN_ROWS, 1000))]
    if state.scalars[S_BELL]:
        cases += [("blocked-ELL width", lambda b: b.scalars.__setitem__(S_BELL_WIDTH, b.scalars[S_BELL_WIDTH] + 1)),
                  ("blocked-ELL values", lambda b: b.bytes.__setitem__(10, b.bytes[10] // 2))]
    if state.scalars[S_GROUPS_VALID]:
        cases += [("row groups", lambda b: b.scalars.__setitem__(S_NGROUPS, b.scalars[S_NGROUPS] + 5))]
    assert which != "laplace" or len(cases) >= 6
    assert which != "block_dense" or len(cases) >= 6
    for name, fn in cases:
        st, none = mutated(fn)
        assert st == 5 and none is None, (which, name, st)  # aoclsparse_status_invalid_value
    st, ok = mutated(lambda b: None)  # (the unmutated state still adopts)
    assert st == 0 and ok is not None


def test_library_rccl_communicator_one_rank():
    """aoclsparse_mi355_comm_*: librccl.so is loaded with dlopen, a communicator of ONE rank is created on the library's device
    (all a one-GPU box can do: RCCL refuses two ranks on one GPU), and every collective the multi-GPU job uses runs through it --
    ncclBroadcast of a handle's analysed state (root side), ncclAllGather, ncclBroadcast of a buffer.  In a child process: the
    communicator is process-wide."""
    code = textwrap.dedent("""
        import sys, ctypes, numpy as np, torch
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        from util import pkg, laplace5
        P = pkg(); L = P.lib()
        assert L.aoclsparse_mi355_comm_info(None, None, None) != 0          # no communicator yet
        assert L.aoclsparse_mi355_comm_allgather(None, None, 0) != 0
        cid = P.CommId()
        assert L.aoclsparse_mi355_comm_unique_id(cid) == 0
        assert L.aoclsparse_mi355_comm_init(1, 0, cid) == 0
        assert L.aoclsparse_mi355_comm_init(1, 0, cid) != 0                 # one communicator per process
        w, r, ver = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
        assert L.aoclsparse_mi355_comm_info(w, r, ver) == 0 and (w.value, r.value) == (1, 0) and ver.value > 20000
        m, rp, ci, v = laplace5(130)
        A = P.Matrix(0, m, m, rp, ci, v); d = P.Descr()
        assert L.aoclsparse_set_mm_hint(A.h, P.OP_NONE, d.h, 10) == 0 and L.aoclsparse_optimize(A.h) == 0
        h = ctypes.c_void_p(A.h.value)
        assert L.aoclsparse_mi355_comm_broadcast_matrix(ctypes.byref(h), 0) == 0 and h.value == A.h.value
        none = ctypes.c_void_p()
        assert L.aoclsparse_mi355_comm_broadcast_matrix(ctypes.byref(none), 0) != 0   # the root must pass its handle
        assert L.aoclsparse_mi355_comm_broadcast_matrix(ctypes.byref(h), 1) != 0      # no such rank
        src = torch.arange(4096, dtype=torch.float64, device="cuda"); dst = torch.zeros_like(src)
        assert L.aoclsparse_mi355_comm_allgather(src.data_ptr(), dst.data_ptr(), src.numel() * 8) == 0
        assert L.aoclsparse_mi355_comm_broadcast(src.data_ptr(), src.numel() * 8, 0) == 0
        assert L.aoclsparse_mi355_synchronize() == 0 and torch.equal(src, dst)
        # the handle still works after having been the root of a broadcast
        n = 8; B = np.ones(m * n); C = np.zeros(m * n)
        assert P.dcsrmm(P.OP_NONE, 1.0, A, d, P.ORDER_COLUMN, B, n, m, 0.0, C, m) == 0
        assert L.aoclsparse_mi355_comm_destroy() == 0 and L.aoclsparse_mi355_comm_info(None, None, None) != 0
        print("rccl", ver.value)
    """) % (HERE, ROOT)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"), capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and "rccl" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])


# --------------------------------------------------------------------------------------------------
# raw aoclsparse_dcsrmv on device arrays: the cached plan is validated INSIDE the product kernel
# --------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("avg", [8, 12])
def test_raw_dcsrmv_stale_plan_is_caught_inside_the_kernel(avg):
    """Different matrices of equal m and nnz written one after the other into the SAME device buffers (a caching allocator after
    free + malloc).  The cache key (row_ptr address, m, nnz, base) hits, the cached block table is stale: every workgroup of
    csr_adaptive_kernel checks its own entry against the live row_ptr and, on a mismatch, computes its rows from the live arrays
    -- the FIRST product after the switch must already be the new matrix's, bit for bit, with no stream round trip; the next call
    sees the raised word and rebuilds.  avg = 8: scalar order (nnz <= 10 m); avg = 12: the 8-lane order (nnz > 10 m,
    csrmv_avx512.cpp:36-134).  The stale cases include a live row far longer than the LDS tile inside a planned multi-row block
    and planned single-row blocks whose live rows are short."""
    m = n = 30000
    rng = np.random.default_rng(40 + avg)
    total = avg * m

    def build(lens):
        lens = np.asarray(lens, np.int64)
        assert lens.sum() == total and lens.max() <= n
        rp = np.zeros(m + 1, np.int32)
        rp[1:] = np.cumsum(lens)
        ci = np.concatenate([np.sort(rng.choice(n, size=int(k), replace=False)) for k in lens]).astype(np.int32)
        return rp, ci, rng.uniform(-1, 1, len(ci))

    uniform = np.full(m, avg)
    skew = np.full(m, avg)
    skew[: m // 2] -= avg - 2
    skew[m // 2:] += avg - 2
    long_rows = np.full(m, avg)
    long_rows[[7, 12345, m - 1]] += np.array([2700, 1100, 5000])
    take = long_rows.sum() - total
    idx = np.arange(100, 100 + take)  # one entry less on `take` other rows
    long_rows[idx] -= 1
    mats = [build(uniform), build(skew), build(long_rows), build(uniform[::-1]), build(long_rows[::-1].copy())]
    d = P.Descr()
    x = rng.uniform(-1, 1, n)
    xd = dev(x)
    d_rp, d_ci, d_v = dev(mats[0][0]), dev(mats[0][1]), dev(mats[0][2])
    L.aoclsparse_mi355_set_pointer_mode(P.PTR_DEVICE)
    try:
        for order in ([0, 1, 2, 3, 4], [2, 0, 4, 1, 2]):
            for k in order:
                rp, ci, v = mats[k]
                d_rp.copy_(torch.from_numpy(rp)), d_ci.copy_(torch.from_numpy(ci)), d_v.copy_(torch.from_numpy(v))
                torch.cuda.synchronize()
                so, yr = oracle.dcsrmv(-1, 0, 1.0, m, total, v, ci, rp, x, 0.0, np.zeros(m))
                assert so == 0
                for call in range(3):  # 1st: stale plan caught in the kernel; 2nd: rebuilt; 3rd: validated hit
                    yd = torch.full((m,), float("nan"), dtype=torch.float64, device="cuda")
                    assert P.dcsrmv(P.OP_NONE, 1.0, m, n, total, d_v, d_ci, d_rp, d, xd, 0.0, yd) == 0
                    torch.cuda.synchronize()
                    got = yd.cpu().numpy()
                    lens = np.diff(rp)
                    exact = lens <= 512  # rows longer than one LDS tile: tolerance only in the automatic mode (wavefront tree)
                    assert np.array_equal(got[exact], yr[exact]), (avg, k, call)
                    scale = np.add.reduceat(np.abs(v * x[ci]), rp[:-1][lens > 0])
                    err = np.abs(got - yr)[lens > 0]
                    assert np.all(err <= (lens[lens > 0] + 16) * EPS64 * scale), (avg, k, call)
    finally:
        L.aoclsparse_mi355_set_pointer_mode(P.PTR_AUTO)


# --------------------------------------------------------------------------------------------------
# randomized sweeps of the round-4 kernels (seeded; every case bit for bit against the oracle)
# --------------------------------------------------------------------------------------------------
def test_window_and_blocked_kernels_randomized_sweep():
    """Random banded matrices (band offsets, row lengths 0..12, base 0 / 1, m off the workgroup size) with column-major operands --
    the LDS-window kernel where the plan applies, the lane-per-row kernels where it does not -- in both operations (op = T runs
    on the handle's A^T copy and ITS window plan), random column counts / leading dimensions / alpha / beta; and random
    block-dense matrices (block pattern, fill 0.55..1, m and k off the tile size) in both layouts on the MFMA kernels.
    oracle.dcsrmm (csrmm_col_major_ref) is the reference in every case."""
    rng = np.random.default_rng(20260404)
    used_window = used_bell = 0
    for case in range(10):
        m = int(rng.integers(8200, 21000))
        g = int(rng.integers(3, 900))
        offs = sorted(set([0, -1, 1, -g, g] + [int(o) for o in rng.integers(-g, g + 1, size=int(rng.integers(0, 5)))]))
        base = int(rng.integers(0, 2))
        rows, cols = [], []
        lens = np.zeros(m, np.int64)
        keep_p = rng.uniform(0.6, 1.0)
        for o in offs:
            r = np.arange(max(0, -o), min(m, m - o))
            r = r[rng.random(len(r)) < keep_p]
            rows.append(r), cols.append(r + o)
        rows, cols = np.concatenate(rows), np.concatenate(cols)
        order = np.lexsort((cols, rows))
        rows, cols = rows[order], cols[order]
        rp = np.zeros(m + 1, np.int64)
        np.add.at(rp, rows + 1, 1)
        rp = np.cumsum(rp)
        v = rng.uniform(-1, 1, len(cols))
        rp32, ci32 = (rp + base).astype(np.int32), (cols + base).astype(np.int32)
        A = P.Matrix(base, m, m, rp32, ci32, v)
        d = P.Descr(base=base)
        for op, opname in ((P.OP_NONE, "n"), (P.OP_TRANSPOSE, "t")):
            n = int(rng.integers(4, 90))
            ldb, ldc = m + 2 * int(rng.integers(0, 3)), m + int(rng.integers(0, 4))
            alpha, beta = float(rng.choice([1.0, -0.5, 2.25])), float(rng.choice([0.0, 0.0, 1.0, -1.5]))
            B, C0 = rng.uniform(-1, 1, ldb * n), rng.uniform(-1, 1, ldc * n)
            if op == P.OP_NONE:
                so, Cr = oracle.dcsrmm("col", alpha, base, v, ci32, rp32, m, B, n, ldb, beta, C0, ldc)
            else:  # the reference transposes A and runs the same kernel (csrmm.hpp:690-760): the oracle on A^T
                import scipy.sparse as sp
                At = sp.csr_matrix((v, cols, rp), shape=(m, m)).T.tocsr()
                At.sort_indices()
                so, Cr = oracle.dcsrmm("col", alpha, base, At.data, (At.indices + base).astype(np.int32),
                                       (At.indptr + base).astype(np.int32), m, B, n, ldb, beta, C0, ldc)
            assert so == 0
            Cd = dev(C0)
            assert P.dcsrmm(op, alpha, A, d, P.ORDER_COLUMN, dev(B), n, ldb, beta, Cd, ldc) == 0
            torch.cuda.synchronize()
            assert _same_bits(Cd.cpu().numpy(), Cr), (case, opname, m, g, offs, n, ldb, ldc, alpha, beta, base)
        used_window += A.spmv_info().mm_window_rows > 0
    assert used_window >= 5, used_window
    for case in range(6):
        nx, ny, nz = (int(t) for t in rng.integers(3, 7, size=3))
        keep = float(rng.uniform(0.55, 1.0))
        m0, rp0, ci0, v0 = standins.block_dense(nx, ny, nz, keep=keep, seed=100 + case)
        mm, kk = m0 - int(rng.integers(0, 16)), m0 - int(rng.integers(0, 16))
        rp, ci, v = _submatrix(m0, rp0, ci0, v0, mm, kk)
        A = P.Matrix(0, mm, kk, rp, ci, v)
        d = P.Descr()
        assert L.aoclsparse_set_mm_hint(A.h, P.OP_NONE, d.h, 10) == 0 and L.aoclsparse_optimize(A.h) == 0
        used_bell += A.spmv_info().mm_bell_width > 0
        for layout in ("row", "col"):
            n = int(rng.integers(1, 100))
            alpha, beta = float(rng.choice([1.0, -0.5])), float(rng.choice([0.0, 1.0, -1.5]))
            if layout == "col":
                ldb, ldc = kk + int(rng.integers(0, 4)), mm + int(rng.integers(0, 4))
                B, C0 = rng.uniform(-1, 1, ldb * n), rng.uniform(-1, 1, ldc * n)
                so, Cr = oracle.dcsrmm("col", alpha, 0, v, ci, rp, mm, B, n, ldb, beta, C0, ldc)
                Cd = dev(C0)
                assert P.dcsrmm(P.OP_NONE, alpha, A, d, P.ORDER_COLUMN, dev(B), n, ldb, beta, Cd, ldc) == 0
                torch.cuda.synchronize()
                assert _same_bits(Cd.cpu().numpy(), Cr), (case, layout, mm, kk, n, ldb, ldc, alpha, beta, keep)
            else:
                ldb, ldc = n + int(rng.integers(0, 3)), n + int(rng.integers(0, 3))
                B, C0 = rng.uniform(-1, 1, kk * ldb), rng.uniform(-1, 1, mm * ldc)
                Bc = np.ascontiguousarray(B.reshape(kk, ldb)[:, :n].T).ravel()
                Cc = np.ascontiguousarray(C0.reshape(mm, ldc)[:, :n].T).ravel()
                so, Cr = oracle.dcsrmm("col", alpha, 0, v, ci, rp, mm, Bc, n, kk, beta, Cc, mm)
                Cd = dev(C0)
                assert P.dcsrmm(P.OP_NONE, alpha, A, d, P.ORDER_ROW, dev(B), n, ldb, beta, Cd, ldc) == 0
                torch.cuda.synchronize()
                got = Cd.cpu().numpy().reshape(mm, ldc)
                assert _same_bits(got[:, :n], Cr.reshape(n, mm).T), (case, layout, mm, kk, n, ldb, ldc, alpha, beta, keep)
                assert np.array_equal(got[:, n:], C0.reshape(mm, ldc)[:, n:])
    assert used_bell >= 4, used_bell


def test_float_short_row_kernel_four_slices_per_wavefront():
    """sell_mv_short_kernel<float, W, 4, SHARED, 4> (round 4: float launches of >= 100,000 slices walk four slices per wavefront):
    a 2,600^2 stencil (shared column lists; 105,625 slices, the last wavefront's group is partial) and a random matrix of 6.5 M + 21
    rows with <= 5 entries and a band of empty rows (own lists, slices narrower than the widest, a partial last slice) -- bit for
    bit against the reference's float order (8 lanes: the scalar chain for rows of < 8 entries), alpha / beta, NaN in x."""
    import __graft_entry__ as entry
    rng = np.random.default_rng(11)
    ml, rpl, cil, vl = entry.laplace5(2600)
    m = 6500000 + 21
    lens = np.where(rng.random(m) < 0.85, 5, rng.integers(0, 6, m)).astype(np.int64)
    lens[200000:200300] = 0
    rp = np.zeros(m + 1, dtype=np.int64)
    np.cumsum(lens, out=rp[1:])
    nnz = int(rp[m])
    rows = np.repeat(np.arange(m, dtype=np.int64), lens)
    # distinct sorted columns per row: a random start + strictly increasing random steps (wrapped rows are re-sorted below)
    steps = rng.integers(1, 40000, nnz).astype(np.int64)
    within = np.arange(nnz, dtype=np.int64) - rp[rows]
    cs = np.cumsum(steps)
    first = cs[rp[:-1].clip(max=nnz - 1)][rows] - steps[rp[:-1].clip(max=nnz - 1)][rows]
    ci = (rng.integers(0, m - 250000, m)[rows] + (cs - first - steps * (within == 0))) % m
    ci = ci.astype(np.int32)
    assert ci.min() >= 0 and ci.max() < m
    key = rows * m + ci
    assert np.all(np.diff(key) > 0)  # sorted and distinct inside every row (starts leave room for 5 steps: no wrap)
    v = rng.uniform(-1, 1, nnz).astype(np.float32)
    for name, mm, rpp, cii, vv in (("stencil", ml, rpl, cil, vl.astype(np.float32)), ("random", m, rp.astype(np.int32), ci, v)):
        A = P.Matrix(0, mm, mm, rpp, cii, vv)
        d = P.Descr()
        assert L.aoclsparse_set_mv_hint(A.h, P.OP_NONE, d.h, 100) == 0 and L.aoclsparse_optimize(A.h) == 0
        inf = A.spmv_info()
        assert inf.kernel in (3, 4) and (mm + 63) // 64 >= 100000, (name, inf.kernel)
        x = rng.uniform(-1, 1, mm).astype(np.float32)
        x[123457] = np.nan
        y0 = rng.uniform(-1, 1, mm).astype(np.float32)
        for alpha, beta in ((1.0, 0.0), (-0.5, 1.25)):
            y = y0.copy()
            assert P.smv(P.OP_NONE, alpha, A, d, x, beta, y) == 0
            st, yr = oracle.scsrmv("lane8", 0, alpha, mm, vv, cii, rpp, x, beta, y0.copy())
            assert st == 0
            assert np.array_equal(np.isnan(y), np.isnan(yr)), name
            ok = ~np.isnan(yr)
            assert np.array_equal(y[ok], yr[ok]), (name, alpha, beta)
        del A


def test_float_plans_use_2048_entry_row_blocks():
    """Large float matrices plan 2,048-entry row blocks (round 4: matrix.cpp choose_tile).  On a 1,000^2 stencil (5 M non-zeros)
    and a random matrix with rows of up to 700 entries (blocks of one to hundreds of rows, rows that do not fit the rest of a
    block): raw aoclsparse_scsrmv on device arrays and aoclsparse_smv on a handle without a SELL copy, bit for bit against the
    reference's float order; the 32-column row-major slab (csrmm_tile_kernel over the same blocks) equals the same columns of a
    160-column product (another kernel, the same chains).  A double handle of the same matrix keeps 1,024, and so does a float
    handle with an mm hint."""
    import __graft_entry__ as entry
    rng = np.random.default_rng(12)
    ml, rpl, cil, vl = entry.laplace5(1000)
    mr = 140000
    rpr, cir, vr = random_csr(5, mr, mr, lambda r, i: int(r.integers(300, 700)) if i % 97 == 0 else int(r.integers(0, 60)))
    assert len(vr) >= 1024 * 256 * 16
    d = P.Descr()
    assert L.aoclsparse_mi355_set_option(P.OPTION_SELL, 0) == 0
    try:
        for name, m, rp, ci, v in (("stencil", ml, rpl, cil, vl), ("random", mr, rpr, cir, vr)):
            vf = v.astype(np.float32)
            A = P.Matrix(0, m, m, rp, ci, vf)
            assert L.aoclsparse_set_mv_hint(A.h, P.OP_NONE, d.h, 100) == 0 and L.aoclsparse_optimize(A.h) == 0
            x = rng.uniform(-1, 1, m).astype(np.float32)
            y0 = rng.uniform(-1, 1, m).astype(np.float32)
            y = y0.copy()
            assert P.smv(P.OP_NONE, -0.5, A, d, x, 1.25, y) == 0
            inf = A.spmv_info()
            assert inf.kernel == 1 and inf.tile == 2048, (name, inf.kernel, inf.tile)
            st, yr = oracle.scsrmv("lane8", 0, -0.5, m, vf, ci, rp, x, 1.25, y0.copy())
            assert st == 0 and np.array_equal(y, yr), name
            L.aoclsparse_mi355_set_pointer_mode(P.PTR_DEVICE)
            try:
                yd = dev(y0)
                assert P.scsrmv(P.OP_NONE, -0.5, m, m, len(vf), dev(vf), dev(ci), dev(rp), d, dev(x), 1.25, yd) == 0
                torch.cuda.synchronize()
                assert np.array_equal(yd.cpu().numpy(), yr), name
            finally:
                L.aoclsparse_mi355_set_pointer_mode(P.PTR_AUTO)
            n = 160
            B = rng.uniform(-1, 1, (m, n)).astype(np.float32)
            C0 = rng.uniform(-1, 1, (m, n)).astype(np.float32)
            Cw = dev(C0.ravel())
            assert P.scsrmm(P.OP_NONE, 1.5, A, d, P.ORDER_ROW, dev(B.ravel()), n, n, -0.5, Cw, n) == 0
            for j0 in (0, 64):
                Bs, Cs0 = np.ascontiguousarray(B[:, j0:j0 + 32]), np.ascontiguousarray(C0[:, j0:j0 + 32])
                Cs = dev(Cs0.ravel())
                assert P.scsrmm(P.OP_NONE, 1.5, A, d, P.ORDER_ROW, dev(Bs.ravel()), 32, 32, -0.5, Cs, 32) == 0
                torch.cuda.synchronize()
                assert np.array_equal(Cs.cpu().numpy().reshape(m, 32), Cw.cpu().numpy().reshape(m, n)[:, j0:j0 + 32]), (name, j0)
            Ad = P.Matrix(0, m, m, rp, ci, v)
            assert L.aoclsparse_set_mv_hint(Ad.h, P.OP_NONE, d.h, 100) == 0 and L.aoclsparse_optimize(Ad.h) == 0
            assert Ad.spmv_info().tile == 1024
            # ... and so does a float handle that announces csrmm (the slab kernel walks the same blocks and prefers 1,024)
            Am = P.Matrix(0, m, m, rp, ci, vf)
            assert L.aoclsparse_set_mm_hint(Am.h, P.OP_NONE, d.h, 100) == 0 and L.aoclsparse_optimize(Am.h) == 0
            assert Am.spmv_info().tile == 1024
    finally:
        assert L.aoclsparse_mi355_set_option(P.OPTION_SELL, -1) == 0


def _export_csr(h, double=True):
    base, m, n, nnz = ctypes.c_int(), ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
    rp, ci, v = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
    fn = L.aoclsparse_export_dcsr if double else L.aoclsparse_export_scsr
    assert fn(h, ctypes.byref(base), ctypes.byref(m), ctypes.byref(n), ctypes.byref(nnz), ctypes.byref(rp), ctypes.byref(ci),
              ctypes.byref(v)) == 0
    k = max(nnz.value, 1)
    row = np.ctypeslib.as_array(ctypes.cast(rp, ctypes.POINTER(ctypes.c_int32)), (m.value + 1,)).copy()
    col = np.ctypeslib.as_array(ctypes.cast(ci, ctypes.POINTER(ctypes.c_int32)), (k,))[: nnz.value].copy()
    val = np.ctypeslib.as_array(ctypes.cast(v, ctypes.POINTER(ctypes.c_double if double else ctypes.c_float)), (k,))[: nnz.value].copy()
    return m.value, n.value, nnz.value, row, col, val


def _spgemm_operands(seed, m, k, n, heavy):
    """A (m x k) and B (k x n) that reach every bin of spgemm_hash_kernel: rows of 0-8 entries (lists of <= 32), of 20-60 (<= 256 /
    <= 2,048), and -- heavy -- a few rows of A with thousands of entries so that their lists exceed 2,048 entries and their upper
    bounds 8,192 (tables in the global slab).  Some B rows are unsorted, some repeat a column (legal input: csr_util.cpp:244)."""
    rng = np.random.default_rng(seed)

    def rows(nr, nc, lens, shuffle_every, dup_every):
        ptr, ind = [0], []
        for i in range(nr):
            c = np.sort(rng.choice(nc, size=min(int(lens[i]), nc), replace=False))
            if shuffle_every and i % shuffle_every == 1 and len(c) > 2:
                c = rng.permutation(c)
            if dup_every and i % dup_every == 2 and len(c) > 3:
                c = c.copy()
                c[len(c) // 2] = c[0] if c[0] != i else c[1]  # a repeated off-diagonal column, out of order
            ind.append(c)
            ptr.append(ptr[-1] + len(c))
        ind = np.concatenate(ind).astype(np.int32) if ptr[-1] else np.zeros(0, np.int32)
        return np.array(ptr, np.int32), ind, rng.uniform(-1, 1, len(ind))

    la = np.where(rng.random(m) < 0.5, rng.integers(0, 9, m), rng.integers(20, 61, m))
    lb = np.where(rng.random(k) < 0.5, rng.integers(0, 9, k), rng.integers(20, 61, k))
    if heavy:
        la[[5, m // 2, m - 3]] = [2500, 4000, 900]
        lb[[7, k - 1]] = [2900, 1500]
    pa, ia, va = rows(m, k, la, 0, 0)
    pb, ib, vb = rows(k, n, lb, 13, 17)
    return (pa, ia, va), (pb, ib, vb)


@pytest.mark.parametrize("heavy", [False, True])
def test_spgemm_hash_kernel_every_bin_bit_exact(heavy):
    """C = A * B through aoclsparse_sp2m: row_ptr, col_ind (first-touch order) and val bit for bit those of the reference's
    two-stage Gustavson (oracle), with rows in every bin of spgemm_hash_kernel, unsorted B rows and repeated columns; full
    computation and the two-stage protocol (finalize twice: the second fill replaces the first); float through ?csr2m."""
    m, k, n = 4000, 5000, 6000
    (pa, ia, va), (pb, ib, vb) = _spgemm_operands(31 + heavy, m, k, n, heavy)
    A, B = P.Matrix(0, m, k, pa, ia, va), P.Matrix(0, k, n, pb, ib, vb)
    assert A.status == 0 and B.status == 0
    d = P.Descr()
    so, pc, ic, vc = oracle.dcsr2m(m, n, 0, pa, ia, va, 0, pb, ib, vb)
    assert so == 0
    counts = np.diff(pc)
    assert counts.min() == 0 and (counts > 32).any() and (counts > 256).any()
    if heavy:
        assert (counts > 2048).sum() >= 3
    C = ctypes.c_void_p()
    assert L.aoclsparse_sp2m(P.OP_NONE, d.h, A.h, P.OP_NONE, d.h, B.h, P.STAGE_FULL, ctypes.byref(C)) == 0
    cm, cn, cz, row, col, val = _export_csr(C)
    assert (cm, cn, cz) == (m, n, len(ic))
    assert np.array_equal(row, pc) and np.array_equal(col, ic) and np.array_equal(val, vc)
    assert L.aoclsparse_destroy(ctypes.byref(C)) == 0
    C = ctypes.c_void_p()
    assert L.aoclsparse_sp2m(P.OP_NONE, d.h, A.h, P.OP_NONE, d.h, B.h, P.STAGE_NNZ_COUNT, ctypes.byref(C)) == 0
    _, _, cz, row, _, _ = _export_csr(C)
    assert cz == len(ic) and np.array_equal(row, pc)
    for _ in range(2):
        assert L.aoclsparse_sp2m(P.OP_NONE, d.h, A.h, P.OP_NONE, d.h, B.h, P.STAGE_FINALIZE, ctypes.byref(C)) == 0
        _, _, _, row, col, val = _export_csr(C)
        assert np.array_equal(col, ic) and np.array_equal(val, vc)
    assert L.aoclsparse_destroy(ctypes.byref(C)) == 0
    Af, Bf = P.Matrix(0, m, k, pa, ia, va.astype(np.float32)), P.Matrix(0, k, n, pb, ib, vb.astype(np.float32))
    C = ctypes.c_void_p()
    assert L.aoclsparse_scsr2m(P.OP_NONE, d.h, Af.h, P.OP_NONE, d.h, Bf.h, P.STAGE_FULL, ctypes.byref(C)) == 0
    _, _, cz, row, col, valf = _export_csr(C, double=False)
    assert cz == len(ic) and np.array_equal(row, pc) and np.array_equal(col, ic)
    scale = np.abs(vc) + 1e-3
    assert np.all(np.abs(valf - vc) <= 4000 * np.finfo(np.float32).eps * np.maximum(scale, 1.0))
    assert L.aoclsparse_destroy(ctypes.byref(C)) == 0


def test_sp2m_result_stays_resident_and_feeds_the_next_product():
    """The handle aoclsparse_sp2m returns holds its CSR arrays in HBM (round 4): ?mv on it and a second product with it start from
    there -- same bits as from the exported host arrays -- and a fill into a handle that has been used drops what it derived."""
    m = 3000
    (pa, ia, va), (pb, ib, vb) = _spgemm_operands(77, m, m, m, False)
    A, B = P.Matrix(0, m, m, pa, ia, va), P.Matrix(0, m, m, pb, ib, vb)
    d = P.Descr()
    C = ctypes.c_void_p()
    assert L.aoclsparse_sp2m(P.OP_NONE, d.h, A.h, P.OP_NONE, d.h, B.h, P.STAGE_FULL, ctypes.byref(C)) == 0
    _, _, cz, row, col, val = _export_csr(C)
    rng = np.random.default_rng(1)
    x, y = rng.uniform(-1, 1, m), np.zeros(m)
    one, zero = ctypes.c_double(1.0), ctypes.c_double(0.0)
    assert L.aoclsparse_dmv(P.OP_NONE, ctypes.byref(one), C, d.h, P._ptr(x), ctypes.byref(zero), P._ptr(y)) == 0
    H = P.Matrix(0, m, m, row, col, val)  # the same matrix from its exported arrays (rows in first-touch order: unsorted)
    y2 = np.zeros(m)
    assert P.dmv(P.OP_NONE, 1.0, H, d, x, 0.0, y2) == 0
    assert np.array_equal(y, y2)
    # C * A with the resident C as the left operand, against the oracle on the exported arrays
    E = ctypes.c_void_p()
    assert L.aoclsparse_sp2m(P.OP_NONE, d.h, C, P.OP_NONE, d.h, A.h, P.STAGE_FULL, ctypes.byref(E)) == 0
    _, _, _, erow, ecol, evalv = _export_csr(E)
    so, pe, ie, ve = oracle.dcsr2m(m, m, 0, row, col, val, 0, pa, ia, va)
    assert so == 0 and np.array_equal(erow, pe) and np.array_equal(ecol, ie) and np.array_equal(evalv, ve)
    assert L.aoclsparse_destroy(ctypes.byref(E)) == 0
    # a second fill of C with other values (A scaled): the products above must not survive in C's device state
    A2 = P.Matrix(0, m, m, pa, ia, 2.0 * va)
    assert L.aoclsparse_sp2m(P.OP_NONE, d.h, A2.h, P.OP_NONE, d.h, B.h, P.STAGE_FINALIZE, ctypes.byref(C)) == 0
    _, _, _, _, col2, val2 = _export_csr(C)
    assert np.array_equal(col2, col) and np.array_equal(val2, 2.0 * val)
    assert L.aoclsparse_dmv(P.OP_NONE, ctypes.byref(one), C, d.h, P._ptr(x), ctypes.byref(zero), P._ptr(y)) == 0
    assert np.array_equal(y, 2.0 * y2)
    assert L.aoclsparse_destroy(ctypes.byref(C)) == 0


@pytest.mark.parametrize("prec", ["z", "c"])
def test_spgemm_hash_kernel_complex_heavy_rows(prec):
    """The same operands as above with complex values (cdouble: 16-byte partial sums, cfloat: 8): the structure -- row_ptr and
    the first-touch column order -- is the real product's, bit for bit; the values are within (terms + 4) eps sum|a||b| of the
    sparse product scipy forms (its own order); rows in every bin including the global-slab one, unsorted and repeated columns."""
    import scipy.sparse as sp
    dtype, eps = (np.complex128, EPS64) if prec == "z" else (np.complex64, np.finfo(np.float32).eps)
    create = L.aoclsparse_create_zcsr if prec == "z" else L.aoclsparse_create_ccsr
    exp = L.aoclsparse_export_zcsr if prec == "z" else L.aoclsparse_export_ccsr
    m, k, n = 4000, 5000, 6000
    (pa, ia, var), (pb, ib, vbr) = _spgemm_operands(32, m, k, n, True)
    rng = np.random.default_rng(5)
    va = (var + 1j * rng.uniform(-1, 1, len(var))).astype(dtype)
    vb = (vbr + 1j * rng.uniform(-1, 1, len(vbr))).astype(dtype)
    hA, hB, C = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
    assert create(ctypes.byref(hA), 0, m, k, len(va), P._ptr(pa), P._ptr(ia), P._ptr(va)) == 0
    assert create(ctypes.byref(hB), 0, k, n, len(vb), P._ptr(pb), P._ptr(ib), P._ptr(vb)) == 0
    d = P.Descr()
    assert L.aoclsparse_sp2m(P.OP_NONE, d.h, hA, P.OP_NONE, d.h, hB, P.STAGE_FULL, ctypes.byref(C)) == 0
    b, m_, n_, z_ = ctypes.c_int(), ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
    a1, a2, a3 = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
    assert exp(C, ctypes.byref(b), ctypes.byref(m_), ctypes.byref(n_), ctypes.byref(z_), ctypes.byref(a1), ctypes.byref(a2), ctypes.byref(a3)) == 0
    nz = z_.value
    row = np.ctypeslib.as_array(ctypes.cast(a1, ctypes.POINTER(ctypes.c_int32)), (m + 1,)).copy()
    col = np.ctypeslib.as_array(ctypes.cast(a2, ctypes.POINTER(ctypes.c_int32)), (nz,)).copy()
    val = np.ctypeslib.as_array(ctypes.cast(a3, ctypes.POINTER(ctypes.c_float if prec == "c" else ctypes.c_double)), (2 * nz,)).view(dtype).copy()
    so, pc, ic, vc = oracle.dcsr2m(m, n, 0, pa, ia, var, 0, pb, ib, vbr)
    assert so == 0 and (np.diff(pc) > 2048).sum() >= 3
    assert (m_.value, n_.value, nz) == (m, n, len(ic)) and np.array_equal(row, pc) and np.array_equal(col, ic)
    A64 = sp.csr_matrix((va.astype(np.complex128), ia.copy(), pa.copy()), shape=(m, k))  # (copies: scipy sorts shared arrays in place)
    B64 = sp.csr_matrix((vb.astype(np.complex128), ib.copy(), pb.copy()), shape=(k, n))
    R = (A64 @ B64).tocsr()
    S = (abs(A64) @ abs(B64)).tocsr()  # sum |a||b| per entry of C
    rows = np.repeat(np.arange(m), np.diff(row))
    ref = np.asarray(R[rows, col]).ravel()
    scale = np.asarray(S[rows, col]).ravel().real
    terms = int(np.diff(pa).max())
    assert np.all(np.abs(val.astype(np.complex128) - ref) <= (2 * terms + 8) * eps * scale + 1e-300)
    for h in (C, hA, hB):
        assert L.aoclsparse_destroy(ctypes.byref(h)) == 0


def test_sp2m_transposed_operands_are_kept_by_their_handles():
    """op = T on either side: the transpose (the reference's stable counting sort) and its device copy stay with the handle, so the
    second product sends and transposes nothing -- same bits as the first and as the oracle (csr2csc + csr2m) -- and a value
    change through the API (?set_value) drops them."""
    m, k = 3000, 2600
    (pa, ia, va), _ = _spgemm_operands(55, m, k, k, False)
    (pb, ib, vb), _ = _spgemm_operands(56, m, k, k, False)  # B is m x k too: A^T * B is k x k
    A, B = P.Matrix(0, m, k, pa, ia, va), P.Matrix(0, m, k, pb, ib, vb)
    d = P.Descr()

    def expect(va_):
        st, cp, ri, cv = oracle.dcsr2csc(m, k, len(va_), 0, 0, pa, ia, va_)
        assert st == 0
        so, pc, ic, vc = oracle.dcsr2m(k, k, 0, cp, ri.astype(np.int32), cv, 0, pb, ib, vb)
        assert so == 0
        return pc, ic, vc

    pc, ic, vc = expect(va)
    for _ in range(2):
        C = ctypes.c_void_p()
        assert L.aoclsparse_sp2m(P.OP_TRANSPOSE, d.h, A.h, P.OP_NONE, d.h, B.h, P.STAGE_FULL, ctypes.byref(C)) == 0
        _, _, _, row, col, val = _export_csr(C)
        assert np.array_equal(row, pc) and np.array_equal(col, ic) and np.array_equal(val, vc)
        assert L.aoclsparse_destroy(ctypes.byref(C)) == 0
    # A[i, j] := 9.5 for the first stored entry of row 40 (aoclsparse_dset_value writes the caller's array and drops every copy)
    i, j = 40, int(ia[pa[40]])
    assert L.aoclsparse_dset_value(A.h, i, j, 9.5) == 0
    assert A.val[pa[40]] == 9.5
    pc2, ic2, vc2 = expect(A.val)
    C = ctypes.c_void_p()
    assert L.aoclsparse_sp2m(P.OP_TRANSPOSE, d.h, A.h, P.OP_NONE, d.h, B.h, P.STAGE_FULL, ctypes.byref(C)) == 0
    _, _, _, row, col, val = _export_csr(C)
    assert np.array_equal(row, pc2) and np.array_equal(col, ic2) and np.array_equal(val, vc2) and not np.array_equal(val, vc)
    assert L.aoclsparse_destroy(ctypes.byref(C)) == 0
    # B^T on the right: A * B^T is m x m
    st, cpb, rib, cvb = oracle.dcsr2csc(m, k, len(vb), 0, 0, pb, ib, vb)
    so, pe, ie, ve = oracle.dcsr2m(m, m, 0, pa, ia, A.val, 0, cpb, rib.astype(np.int32), cvb)
    for _ in range(2):
        E = ctypes.c_void_p()
        assert L.aoclsparse_sp2m(P.OP_NONE, d.h, A.h, P.OP_TRANSPOSE, d.h, B.h, P.STAGE_FULL, ctypes.byref(E)) == 0
        _, _, _, row, col, val = _export_csr(E)
        assert so == 0 and np.array_equal(row, pe) and np.array_equal(col, ie) and np.array_equal(val, ve)
        assert L.aoclsparse_destroy(ctypes.byref(E)) == 0


def test_sp2m_finalize_refuses_a_row_ptr_of_another_product():
    """stage_finalize trusts nothing about the handle it is given beyond what it can check: a C whose row_ptr came from stage 1 of a
    DIFFERENT product of the same dimensions (rows that would end short of, or run beyond, their segments; counts above the upper
    bound) is refused with invalid_value -- no row writes into its neighbour's segment, no table fills up -- and the right C still
    finalizes afterwards."""
    m = 3000
    (pa, ia, va), (pb, ib, vb) = _spgemm_operands(91, m, m, m, False)
    (pa2, ia2, va2), _ = _spgemm_operands(92, m, m, m, False)  # another pattern, same shape
    A, A2, B = P.Matrix(0, m, m, pa, ia, va), P.Matrix(0, m, m, pa2, ia2, va2), P.Matrix(0, m, m, pb, ib, vb)
    d = P.Descr()
    C = ctypes.c_void_p()
    assert L.aoclsparse_sp2m(P.OP_NONE, d.h, A.h, P.OP_NONE, d.h, B.h, P.STAGE_NNZ_COUNT, ctypes.byref(C)) == 0
    assert L.aoclsparse_sp2m(P.OP_NONE, d.h, A2.h, P.OP_NONE, d.h, B.h, P.STAGE_FINALIZE, ctypes.byref(C)) == 5  # invalid_value
    assert L.aoclsparse_sp2m(P.OP_NONE, d.h, A.h, P.OP_NONE, d.h, B.h, P.STAGE_FINALIZE, ctypes.byref(C)) == 0
    so, pc, ic, vc = oracle.dcsr2m(m, m, 0, pa, ia, va, 0, pb, ib, vb)
    _, _, _, row, col, val = _export_csr(C)
    assert so == 0 and np.array_equal(row, pc) and np.array_equal(col, ic) and np.array_equal(val, vc)
    # a refused finalize on a COMPLETE result leaves it as it was (round 5: the device verdict is read before the handle is touched)
    assert L.aoclsparse_sp2m(P.OP_NONE, d.h, A2.h, P.OP_NONE, d.h, B.h, P.STAGE_FINALIZE, ctypes.byref(C)) == 5
    _, _, _, row, col, val = _export_csr(C)
    assert np.array_equal(row, pc) and np.array_equal(col, ic) and np.array_equal(val, vc)
    x = np.random.default_rng(5).uniform(-1, 1, m)
    y = np.zeros(m)
    one, zero = ctypes.c_double(1.0), ctypes.c_double(0.0)
    assert L.aoclsparse_dmv(P.OP_NONE, ctypes.byref(one), C, d.h, P._ptr(x), ctypes.byref(zero), P._ptr(y)) == 0
    _, yr = oracle.dcsrmv(-1, 0, 1.0, m, len(vc), vc, ic, pc, x, 0.0, np.zeros(m))  # (the reference's dispatch rule picks the order)
    short = np.diff(pc) < 32
    assert np.array_equal(y[short], yr[short]) and np.allclose(y, yr, rtol=1e-12, atol=1e-12)
    assert L.aoclsparse_destroy(ctypes.byref(C)) == 0


@pytest.mark.parametrize("prec", ["z", "c"])
def test_complex_mv_on_the_sell_copy(prec):
    """aoclsparse_{z,c}mv on a handle with an mv hint runs on the SELL-64 copy (round 4): general (rectangular) x N / T / H,
    symmetric / hermitian / triangular x fill x N / T / H, both bases, alpha / beta classes, within (2 len + 16) eps of the restated
    operator; small launches (general kernel, slices of every width incl. empty rows and rows of 40 entries) and a launch of
    > 4,096 slices with rows of <= 8 entries (short-row kernel; a stencil with shared column lists and a random matrix without);
    beta == 0 never reads y; host and device vectors give the same bits."""
    import scipy.sparse as sp
    import __graft_entry__ as entry
    dtype, eps = (np.complex128, EPS64) if prec == "z" else (np.complex64, np.finfo(np.float32).eps)
    create = L.aoclsparse_create_zcsr if prec == "z" else L.aoclsparse_create_ccsr
    mv = L.aoclsparse_zmv if prec == "z" else L.aoclsparse_cmv
    rng = np.random.default_rng(211)
    alpha, beta = np.array([0.7 - 0.4j], dtype), np.array([-0.3 + 0.2j], dtype)
    zero = np.zeros(1, dtype)
    ops = {"n": P.OP_NONE, "t": P.OP_TRANSPOSE, "h": P.OP_CONJ_TRANSPOSE}

    def cplx_csr(seed, m, n, lens_of):
        r = np.random.default_rng(seed)
        lens = np.minimum(lens_of(r, m), n)
        rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        ci = np.concatenate([np.sort(r.choice(n, int(k), replace=False)) for k in lens] + [np.zeros(0, np.int64)]).astype(np.int32)
        v = (r.uniform(-1, 1, len(ci)) + 1j * r.uniform(-1, 1, len(ci))).astype(dtype)
        return rp, ci, v

    # small: every descriptor (the oracle's dense restatement)
    for base in (0, 1):
        for (m, n, cases) in ((700, 640, [("general", "lower", "non_unit")]),
                              (600, 600, [(t, f, "non_unit") for t in ("symmetric", "hermitian", "triangular") for f in ("lower", "upper")])):
            rp, ci, v = cplx_csr(300 + base, m, n, lambda r, mm: np.where(r.random(mm) < 0.1, 40, r.integers(0, 12, mm)))
            if m == n:  # a full, real diagonal (hermitian)
                A0 = sp.csr_matrix((v, ci, rp), shape=(m, n)).tolil()
                A0.setdiag(3.0)
                A0 = A0.tocsr(); A0.sort_indices()
                rp, ci, v = A0.indptr.astype(np.int32), A0.indices.astype(np.int32), A0.data.astype(dtype)
            rpb, cib = rp + base, ci + base
            h = ctypes.c_void_p()
            assert create(ctypes.byref(h), base, m, n, len(v), P._ptr(rpb), P._ptr(cib), P._ptr(v)) == 0
            lens = np.diff(rp)
            for mtype, fill, diag in cases:
                d = P.Descr(base=base, mtype={"general": 0, "symmetric": 1, "hermitian": 2, "triangular": 3}[mtype],
                            fill=P.FILL_LOWER if fill == "lower" else P.FILL_UPPER)
                for opn, op in ops.items():
                    assert L.aoclsparse_set_mv_hint(h, op, d.h, 10) == 0
                    nx, ny = (n, m) if opn == "n" or mtype != "general" else (m, n)
                    x = (rng.uniform(-1, 1, nx) + 1j * rng.uniform(-1, 1, nx)).astype(dtype)
                    y0 = (rng.uniform(-1, 1, ny) + 1j * rng.uniform(-1, 1, ny)).astype(dtype)
                    yr, scale = oracle.zmv(opn, mtype, fill, diag, base, alpha[0], m, n, rpb, cib, v, x, beta[0], y0)
                    y = y0.copy()
                    assert mv(op, P._ptr(alpha), h, d.h, P._ptr(x), P._ptr(beta), P._ptr(y)) == 0, (mtype, opn)
                    bound = (2 * max(lens.max(), 1) * (2 if mtype in ("symmetric", "hermitian") else 1) + 16) * eps * (scale + 1e-30)
                    assert np.all(np.abs(y - yr) <= bound), (prec, base, mtype, fill, opn, float(np.max(np.abs(y - yr) / bound)))
                    yd = dev(y0)
                    assert mv(op, P._ptr(alpha), h, d.h, P._ptr(dev(x)), P._ptr(beta), P._ptr(yd)) == 0
                    torch.cuda.synchronize()
                    assert np.array_equal(yd.cpu().numpy(), y)
            if m != n:
                assert P.Matrix.from_handle  # (binding present)
            ynan = np.full(m, np.nan + 1j * np.nan, dtype)
            xx = rng.uniform(-1, 1, n).astype(dtype)
            assert mv(P.OP_NONE, P._ptr(alpha), h, P.Descr(base=base).h, P._ptr(xx), P._ptr(zero), P._ptr(ynan)) == 0
            assert not np.any(np.isnan(ynan))
            L.aoclsparse_destroy(ctypes.byref(h))
    # large launches: the short-row kernel (> 4,096 slices, widest slice <= 8), shared lists (stencil) and own lists (random)
    ml, rpl, cil, vl = entry.laplace5(560)
    rpr, cir, vr = cplx_csr(77, 300037, 300037, lambda r, mm: np.where(r.random(mm) < 0.85, 7, r.integers(0, 8, mm)))
    vl = (vl * (1.0 + 0.5j) + 0.25j * np.cos(np.arange(len(vl)))).astype(dtype)
    for name, m, rp, ci, v in (("stencil", ml, rpl, cil, vl), ("random", 300037, rpr, cir, vr)):
        h = ctypes.c_void_p()
        assert create(ctypes.byref(h), 0, m, m, len(v), P._ptr(rp), P._ptr(ci), P._ptr(v)) == 0
        d = P.Descr()
        A64 = sp.csr_matrix((v.astype(np.complex128), ci, rp), shape=(m, m))
        Aabs = abs(A64)
        x = (rng.uniform(-1, 1, m) + 1j * rng.uniform(-1, 1, m)).astype(dtype)
        y0 = (rng.uniform(-1, 1, m) + 1j * rng.uniform(-1, 1, m)).astype(dtype)
        for opn, op in ops.items():
            assert L.aoclsparse_set_mv_hint(h, op, d.h, 10) == 0
            Mo = {"n": A64, "t": A64.T, "h": A64.conj().T}[opn]
            Mabs = Aabs if opn == "n" else Aabs.T
            x64, y64 = x.astype(np.complex128), y0.astype(np.complex128)
            yr = alpha[0] * (Mo @ x64) + beta[0] * y64
            scale = abs(alpha[0]) * (Mabs @ np.abs(x64)) + abs(beta[0]) * np.abs(y64)
            y = y0.copy()
            assert mv(op, P._ptr(alpha), h, d.h, P._ptr(x), P._ptr(beta), P._ptr(y)) == 0, (name, opn)
            assert np.all(np.abs(y - yr) <= (2 * 8 + 16) * eps * (scale + 1e-30)), (prec, name, opn)
        L.aoclsparse_destroy(ctypes.byref(h))


def test_float_csrmm_four_columns_per_lane_same_bits_as_two():
    """scsrmm row-major, C read, n >= 256 and a multiple of 4 (csrmm_row_wave_rc4_kernel, round 4): every 128-column half of the
    result equals, bit for bit, the product of the same 128 columns through the two-column kernel (n = 128 takes it); padded
    leading dimensions, alpha / beta classes, rows of 0 .. 40 entries, and within 2 (len + 2) eps of the double product."""
    import scipy.sparse as sp
    rng = np.random.default_rng(17)
    m = k = 20000
    rp, ci, v = random_csr(19, m, k, lambda r, i: 0 if i % 50 == 7 else (40 if i % 33 == 0 else r.integers(1, 9)))
    vf = v.astype(np.float32)
    A = P.Matrix(0, m, k, rp, ci, vf)
    d = P.Descr()
    A64 = sp.csr_matrix((vf.astype(np.float64), ci, rp), shape=(m, k))
    lens = np.diff(rp)
    for n, pad, alpha, beta in ((256, 0, 1.0, 0.0), (260, 4, -0.5, 1.25), (512, 8, 2.0, -1.0)):
        ldb = ldc = n + pad
        B = rng.uniform(-1, 1, (k, ldb)).astype(np.float32)
        C0 = rng.uniform(-1, 1, (m, ldc)).astype(np.float32)
        Cw = dev(C0.ravel())
        assert P.scsrmm(P.OP_NONE, alpha, A, d, P.ORDER_ROW, dev(B.ravel()), n, ldb, beta, Cw, ldc) == 0
        torch.cuda.synchronize()
        W = Cw.cpu().numpy().reshape(m, ldc)
        assert np.array_equal(W[:, n:], C0[:, n:])  # the padding columns of C are untouched
        for j0 in range(0, n - 127, 128):
            Bs, Cs0 = np.ascontiguousarray(B[:, j0:j0 + 128]), np.ascontiguousarray(C0[:, j0:j0 + 128])
            Cs = dev(Cs0.ravel())
            assert P.scsrmm(P.OP_NONE, alpha, A, d, P.ORDER_ROW, dev(Bs.ravel()), 128, 128, beta, Cs, 128) == 0
            torch.cuda.synchronize()
            assert np.array_equal(Cs.cpu().numpy().reshape(m, 128), W[:, j0:j0 + 128]), (n, j0)
        ref = alpha * (A64 @ B[:, :n].astype(np.float64)) + beta * C0[:, :n].astype(np.float64)
        scale = abs(alpha) * (abs(A64) @ np.abs(B[:, :n]).astype(np.float64)) + abs(beta) * np.abs(C0[:, :n])
        bound = (2 * (lens.max() + 2)) * np.finfo(np.float32).eps * (scale + 1e-30)
        assert np.all(np.abs(W[:, :n] - ref) <= bound)


def _big_rect_csr(seed, m, n, base, kind):
    """>= 1 M entries, vectorised: rows of 0 .. 14 entries (kind 0: sorted, distinct; 1: + unsorted rows and repeated columns;
    2: + columns of hundreds of entries; 3: + one column of > 2,048 entries, which the device sort declines)."""
    rng = np.random.default_rng(seed)
    lens = rng.integers(0, 15, m).astype(np.int64)
    lens[::97] = 0
    ptr = np.zeros(m + 1, np.int64); np.cumsum(lens, out=ptr[1:])
    nnz = int(ptr[m])
    rows = np.repeat(np.arange(m, dtype=np.int64), lens)
    within = np.arange(nnz, dtype=np.int64) - ptr[rows]
    start = rng.integers(0, n - 15 * 400, m)[rows]
    ind = start + within * rng.integers(1, 400, m)[rows]  # distinct, ascending inside a row
    if kind >= 2:  # hub columns: every 5th row's first entry goes to one of 40 columns -> segments of hundreds
        first = (within == 0) & (rows % 5 == 0)
        hub = (rows[first] // 5) % 40
        ind[first] = np.minimum(hub, start[first])  # (stays the row's smallest column or a repeat of it)
    if kind >= 3:
        first = (within == 0) & (rows % 3 == 1)
        ind[first] = 0
    if kind >= 1:
        sw = (within == 1) & (rows % 11 == 3)  # swap the first two entries of some rows: unsorted
        idx = np.flatnonzero(sw)
        ind[idx], ind[idx - 1] = ind[idx - 1].copy(), ind[idx].copy()
        dup = (within == 2) & (rows % 13 == 5)  # repeat the row's first column (not the diagonal of a square matrix: m != n here)
        idx = np.flatnonzero(dup)
        ind[idx] = ind[idx - 2]
    val = rng.uniform(-1, 1, nnz)
    return (ptr + base).astype(np.int32), (ind + base).astype(np.int32), val


@pytest.mark.parametrize("kind", [0, 1, 2, 3])
def test_csr2csc_large_on_the_device_bit_exact(kind):
    """aoclsparse_dcsr2csc / scsr2csc on >= 1 M entries run the stable counting sort on the device (round 4): col_ptr, row_ind and
    val bit for bit those of the reference's host loop (oracle), for both base pairs that differ, rectangular shapes, empty rows
    and columns, unsorted rows, repeated columns, columns of hundreds of entries (the wavefront sort) and a column of > 2,048
    (declined: the host loop)."""
    m, n = 120000, 150000
    for base_in, base_out in ((0, 1), (1, 0)):
        rp, ci, v = _big_rect_csr(40 + kind, m, n, base_in, kind)
        nnz = len(v)
        assert nnz >= 1 << 19
        while nnz < (1 << 20):  # (the generator's mean row length gives ~0.84 M: two copies stacked keep it above the threshold)
            rp = np.concatenate([rp, rp[1:] + (rp[-1] - base_in)]).astype(np.int32)
            ci = np.concatenate([ci, ci]); v = np.concatenate([v, 2.0 * v])
            nnz = len(v)
        mm = len(rp) - 1
        d = P.Descr(base=base_in)
        st, cp, ri, cv = oracle.dcsr2csc(mm, n, nnz, base_in, base_out, rp, ci, v)
        assert st == 0
        op, oi, ov = np.zeros(n + 1, np.int32), np.zeros(nnz, np.int32), np.zeros(nnz)
        assert L.aoclsparse_dcsr2csc(mm, n, nnz, d.h, base_out, P._ptr(rp), P._ptr(ci), P._ptr(v), P._ptr(oi), P._ptr(op), P._ptr(ov)) == 0
        assert np.array_equal(op, cp) and np.array_equal(oi, ri) and np.array_equal(ov, cv), (kind, base_in)
        if kind == 2:
            seg = np.diff(cp)
            assert seg.max() > 256 and seg.max() <= 2048
        if kind == 3:
            assert np.diff(cp).max() > 2048
        vf = v.astype(np.float32)
        ovf = np.zeros(nnz, np.float32)
        assert L.aoclsparse_scsr2csc(mm, n, nnz, d.h, base_out, P._ptr(rp), P._ptr(ci), P._ptr(vf), P._ptr(oi), P._ptr(op), P._ptr(ovf)) == 0
        assert np.array_equal(op, cp) and np.array_equal(oi, ri) and np.array_equal(ovf, cv.astype(np.float32))


def test_handle_transpose_built_on_the_device():
    """The transposes handles keep (?mv / sp2m with op = T) come from the same device sort for >= 1 M entries: dmv with op = T on a
    rectangular matrix within the bound of the restated product, and sp2m(A^T, B) bit for bit the oracle's csr2csc + csr2m."""
    import scipy.sparse as sp
    m, n = 120000, 150000
    rp, ci, v = _big_rect_csr(61, m, n, 0, 2)
    while len(v) < (1 << 20):
        rp = np.concatenate([rp, rp[1:] + rp[-1]]).astype(np.int32); ci = np.concatenate([ci, ci]); v = np.concatenate([v, 2.0 * v])
    mm = len(rp) - 1
    A = P.Matrix(0, mm, n, rp, ci, v)
    assert A.status == 0
    d = P.Descr()
    rng = np.random.default_rng(2)
    x, y0 = rng.uniform(-1, 1, mm), rng.uniform(-1, 1, n)
    y = y0.copy()
    assert P.dmv(P.OP_TRANSPOSE, 1.5, A, d, x, -0.5, y) == 0
    A64 = sp.csr_matrix((v.copy(), ci.copy(), rp.copy()), shape=(mm, n))  # (copies: scipy canonicalises shared arrays in place)
    ref = 1.5 * (A64.T @ x) - 0.5 * y0
    scale = 1.5 * (abs(A64).T @ np.abs(x)) + 0.5 * np.abs(y0)
    colmax = int(np.bincount(ci, minlength=n).max())
    assert np.all(np.abs(y - ref) <= (2 * colmax + 16) * EPS64 * (scale + 1e-300))
    # sp2m with the kept transpose: A^T (n x mm) times a thin B (mm x 64 columns)
    pb, ib, vb = random_csr(9, mm, 64, lambda r, i: r.integers(0, 3))
    B = P.Matrix(0, mm, 64, pb, ib, vb)
    st, cp, ri, cv = oracle.dcsr2csc(mm, n, len(v), 0, 0, rp, ci, v)
    so, pc, ic, vc = oracle.dcsr2m(n, 64, 0, cp, ri.astype(np.int32), cv, 0, pb, ib, vb)
    C = ctypes.c_void_p()
    assert L.aoclsparse_sp2m(P.OP_TRANSPOSE, d.h, A.h, P.OP_NONE, d.h, B.h, P.STAGE_FULL, ctypes.byref(C)) == 0
    _, _, _, row, col, val = _export_csr(C)
    assert st == 0 and so == 0 and np.array_equal(row, pc) and np.array_equal(col, ic) and np.array_equal(val, vc)
    assert L.aoclsparse_destroy(ctypes.byref(C)) == 0


def test_release_staging_frees_the_scratch_and_the_next_call_allocates_again():
    """aoclsparse_mi355_release_staging: the grow-only scratch a product leaves in HBM (sp2m: a transposed operand of a result
    handle goes through staging slots; dmv with host vectors stages x and y) is freed on request and the next call simply
    allocates it again -- same bits."""
    m, rp, ci, v = laplace5(400)
    A = P.Matrix(0, m, m, rp, ci, v)
    d = P.Descr()
    x, y = np.cos(0.1 * np.arange(m)), np.zeros(m)
    assert P.dmv(P.OP_NONE, 1.0, A, d, x, 0.0, y) == 0  # host vectors: staged
    C = ctypes.c_void_p()
    assert L.aoclsparse_sp2m(P.OP_NONE, d.h, A.h, P.OP_NONE, d.h, A.h, P.STAGE_FULL, ctypes.byref(C)) == 0
    E = ctypes.c_void_p()  # C^T * A: the transpose of a result handle is built per call and staged
    assert L.aoclsparse_sp2m(P.OP_TRANSPOSE, d.h, C, P.OP_NONE, d.h, A.h, P.STAGE_FULL, ctypes.byref(E)) == 0
    _, _, _, row1, col1, val1 = _export_csr(E)
    assert L.aoclsparse_destroy(ctypes.byref(E)) == 0
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    freed = ctypes.c_size_t(0)
    assert L.aoclsparse_mi355_release_staging(ctypes.byref(freed)) == 0
    free1 = torch.cuda.mem_get_info()[0]
    assert freed.value > 8 * m and free1 >= free0
    assert L.aoclsparse_mi355_release_staging(ctypes.byref(freed)) == 0 and freed.value == 0
    y2 = np.zeros(m)
    assert P.dmv(P.OP_NONE, 1.0, A, d, x, 0.0, y2) == 0 and np.array_equal(y, y2)
    assert L.aoclsparse_sp2m(P.OP_TRANSPOSE, d.h, C, P.OP_NONE, d.h, A.h, P.STAGE_FULL, ctypes.byref(E)) == 0
    _, _, _, row2, col2, val2 = _export_csr(E)
    assert np.array_equal(row1, row2) and np.array_equal(col1, col2) and np.array_equal(val1, val2)
    assert L.aoclsparse_destroy(ctypes.byref(E)) == 0 and L.aoclsparse_destroy(ctypes.byref(C)) == 0
