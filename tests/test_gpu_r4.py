"""GPU tier, round 4: argument checks of the column-sharded csrmm entry points on the WHOLE operands, the process-wide beta = 0
mode word, the in-library multi-device bookkeeping (same-device honesty, per-device times), and the kernels added this round.
Every comparison is against the CPU oracle through the C ABI; tolerances are written where they are used."""
import json
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

import oracle
from util import EPS64, ROOT, laplace5, pkg, random_csr

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
