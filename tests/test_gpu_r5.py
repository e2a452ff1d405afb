"""Round-5 GPU tests: the one-launch merge-path kernel under residency pressure and uneven load (tools/stress_mergepath.py in small),
and the differential fuzzer of the round's kernels (tools/fuzz_r5.py) as a short run.  The other round-5 changes are tested where
their subjects already were: test_gpu_parity.py (tree / strict modes, merge-path), test_gpu_configs.py (full-size mix in both modes),
test_gpu_r4.py (blocked-ELL with Inf / NaN, sp2m finalize)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(tool, *args, timeout=600):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool)] + [str(a) for a in args], capture_output=True, text=True,
                       timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])


def test_merge_path_look_back_under_residency_pressure_and_uneven_load():
    """~9,000 tiles of 1,024 items against the 2,048 workgroups the chip holds, three rows that cross ~490 tiles each, ten launches
    (every second one next to a competing stream that keeps all CUs busy): no look-back expires (no NaN), every row within the
    bound, and every launch returns the same bits (the order of additions depends on the tiling only)."""
    out = _run("stress_mergepath.py", 2000000, 3, 500000, 10)
    assert out["tiles"] > 4 * 2048 and out["tiles_per_long_row"] >= 400
    assert out["every_launch_bit_identical"] and out["worst_err_over_bound"] <= 1.0 and out["with_competing_stream"] == 5


def test_round5_differential_fuzz_short():
    """tree / merge-path / strict SpMV, blocked-ELL csrmm with Inf / NaN, device csr2csc with long rows: tools/fuzz_r5.py, 25 cases"""
    out = _run("fuzz_r5.py", 25, 3)
    assert not any(out["mismatches"].values()), out
    assert out["checked"]["spmv"] >= 30 and out["checked"]["bell"] >= 10 and out["checked"]["csr2csc"] >= 2


def test_release_staging_reaches_the_secondary_runtime_slots():
    """aoclsparse_mi355_release_staging is process-wide: the slabs the slots of a multi-device call staged for their
    share of host operands are freed too (here: four slots on device 0), counted in bytes_freed, and the next multi-device call
    allocates them again and returns the same bits."""
    import ctypes
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import pkg, random_csr
    P = pkg()
    L = P.lib()
    m, k, n = 4000, 3000, 64
    rp, ci, v = random_csr(77, m, k, lambda r, i: r.integers(0, 9))
    A = P.Matrix(0, m, k, rp, ci, v)
    d = P.Descr()
    assert L.aoclsparse_set_mm_hint(A.h, P.OP_NONE, d.h, 10) == 0 and L.aoclsparse_optimize(A.h) == 0
    rng = np.random.default_rng(3)
    B, C0 = rng.uniform(-1, 1, k * n), rng.uniform(-1, 1, m * n)
    st, dev0, _, _ = P.device_info()
    assert st == 0
    freed = ctypes.c_size_t(0)
    assert L.aoclsparse_mi355_release_staging(ctypes.byref(freed)) == 0  # start from nothing staged
    C1 = C0.copy()
    assert P.dcsrmm_multi(P.OP_NONE, 1.5, A, d, P.ORDER_ROW, B, n, n, -0.5, C1, n, [dev0] * 4) == 0
    torch.cuda.synchronize()
    assert L.aoclsparse_mi355_release_staging(ctypes.byref(freed)) == 0
    # each of the four slots staged its 16-column slab of B (k x 16) and C (m x 16): at least those bytes come back
    assert freed.value >= 8 * 16 * (k + m) * 3, freed.value
    assert L.aoclsparse_mi355_release_staging(ctypes.byref(freed)) == 0 and freed.value == 0
    C2 = C0.copy()
    assert P.dcsrmm_multi(P.OP_NONE, 1.5, A, d, P.ORDER_ROW, B, n, n, -0.5, C2, n, [dev0] * 4) == 0
    assert np.array_equal(C1, C2)
