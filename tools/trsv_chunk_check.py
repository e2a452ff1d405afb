#!/usr/bin/env python3
"""round 6: the two-level TRSV schedule (5) against the lane-per-block one (4) on the ILU(0) factors of the shell-like stand-ins:
bits against oracle.dtrsv, then the time of 20 solves each.   python3 tools/trsv_chunk_check.py [structured|unstructured|both] [small]
argv[3] = force: build the chunk plan whatever the plan-time model says."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as entry, oracle, standins
pkg = entry.load_package(); L = pkg.lib()
which = sys.argv[1] if len(sys.argv) > 1 else "both"
small = len(sys.argv) > 2 and sys.argv[2] == "small"
if len(sys.argv) > 3 and sys.argv[3] == "force":
    assert L.aoclsparse_mi355_set_option(pkg.OPTION_TRSV_CHUNKS, 1) == 0
dev = torch.device("cuda", 0)
for variant in (("structured", "unstructured") if which == "both" else (which,)):
    n = 150000 if small else 1508065
    m, rp, ci, v = standins.shell_like_unstructured(n=n) if variant == "unstructured" else standins.shell_like(n=n)
    st, lu, dg = oracle.dilu0(m, 0, rp, ci, v)
    o = oracle.dcsr_optimize(m, m, len(lu), 0, rp, ci, lu)
    for fill, unit in ((pkg.FILL_LOWER, True), (pkg.FILL_UPPER, False)):
        A = pkg.Matrix(0, m, m, rp, ci, lu)
        d = pkg.Descr(mtype=pkg.TYPE_TRIANGULAR, fill=fill, diag=pkg.DIAG_UNIT if unit else pkg.DIAG_NON_UNIT)
        b = np.random.default_rng(2).uniform(-1, 1, m)
        for op in (pkg.OP_NONE, pkg.OP_TRANSPOSE):
            key = ("l" if fill == pkg.FILL_LOWER else "u") + ("t" if op == pkg.OP_TRANSPOSE else "")
            so, xr = oracle.dtrsv(key, 1.0, m, 0, lu, ci, rp, o["idiag"] if fill == pkg.FILL_LOWER else o["iurow"], b, unit)
            assert so == 0
            L.aoclsparse_mi355_set_pointer_mode(pkg.PTR_DEVICE)
            bd = torch.from_numpy(b).to(dev)
            out = {"variant": variant, "triangle": key, "unit": unit, "m": m}
            for sched in (4, 5):
                assert L.aoclsparse_mi355_set_trsv_schedule(sched) == 0
                xd = torch.full((m,), float("nan"), dtype=torch.float64, device=dev)
                assert pkg.dtrsv(op, 1.0, A, d, bd, xd) == 0
                torch.cuda.synchronize()
                x = xd.cpu().numpy()
                out["bit_exact_%d" % sched] = bool(np.array_equal(x, xr))
                if not out["bit_exact_%d" % sched]:
                    bad = np.flatnonzero(x != xr)
                    out["first_bad_%d" % sched] = [int(bad[0]), int(len(bad)), float(x[bad[0]]), float(xr[bad[0]])]
                for _ in range(3):
                    pkg.dtrsv(op, 1.0, A, d, bd, xd)
                torch.cuda.synchronize()
                pkg.timer_start()
                for _ in range(20):
                    pkg.dtrsv(op, 1.0, A, d, bd, xd)
                out["ms_%d" % sched] = round(pkg.timer_stop() / 20, 4)
            assert L.aoclsparse_mi355_set_trsv_schedule(-1) == 0
            out["levels"] = A.trsv_levels(fill, op)
            ti = A.trsv_info(fill, op)
            out["plan"] = {"block_levels": ti.block_levels, "chunks": ti.chunks, "steps": ti.steps, "model_two_level_us": ti.model_chunk_us,
                           "model_lane_per_block_us": ti.model_block_us, "automatic_schedule": ti.schedule}
            print(json.dumps(out), flush=True)
        del A
