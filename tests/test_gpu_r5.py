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
