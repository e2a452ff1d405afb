#!/usr/bin/env python3
"""Where does a block level of the supernodal TRSV (schedule 4) spend its time?  Diagnostic: runs one solve of the
shell-like ILU(0) factor with AOCLSPARSE_MI355_TRSV_TRACE set and prints, from the per-slice 100 MHz timestamps the
kernel dumps (after the ticket / dependencies all in / end / block level), one JSON line of statistics."""
import json, os, sys
TRACE = "/tmp/trsv_trace.bin"
os.environ["AOCLSPARSE_MI355_TRSV_TRACE"] = TRACE
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as entry, oracle, standins
pkg = entry.load_package(); L = pkg.lib()
L.aoclsparse_mi355_set_pointer_mode(pkg.PTR_DEVICE)
dev = torch.device("cuda", 0)
m, rp, ci, v = standins.shell_like_unstructured() if os.environ.get("VARIANT") == "unstructured" else standins.shell_like()   # VARIANT=unstructured
st, lu, dg = oracle.dilu0(m, 0, rp, ci, v)
A = pkg.Matrix(0, m, m, rp, ci, lu)
dl = pkg.Descr(mtype=pkg.TYPE_TRIANGULAR, fill=pkg.FILL_LOWER, diag=pkg.DIAG_UNIT)
assert L.aoclsparse_set_sv_hint(A.h, pkg.OP_NONE, dl.h, 100) == 0 and L.aoclsparse_optimize(A.h) == 0
b = np.random.default_rng(2).uniform(-1, 1, m)
bd, xd = torch.from_numpy(b).to(dev), torch.zeros(m, dtype=torch.float64, device=dev)
KID = int(os.environ["KID"]) if "KID" in os.environ else None   # KID=1 / 3: the KT-order block kernel
for _ in range(3):
    pkg.dtrsv(pkg.OP_NONE, 1.0, A, dl, bd, xd, kid=KID)
torch.cuda.synchronize()
t = np.fromfile(TRACE, dtype=np.uint64).reshape(-1, 6).astype(np.int64)
start, ready, done, lev = t[:, 0], t[:, 1], t[:, 2], t[:, 3]
t0 = start.min()
nlev = int(lev.max()) + 1
us = lambda ticks: ticks / 100.0
rd = np.full(nlev, -1, np.int64); dn = np.full(nlev, -1, np.int64); cnt = np.zeros(nlev, np.int64)
rdmin = np.full(nlev, 1 << 62, np.int64)
for s in range(len(lev)):
    l = lev[s]
    rd[l] = max(rd[l], ready[s]); dn[l] = max(dn[l], done[s]); cnt[l] += 1; rdmin[l] = min(rdmin[l], ready[s])
hop = us(rd[1:] - dn[:-1])          # last slice of level l ready, after the last slice of level l-1 finished
work = us(done - ready)             # per slice: dependencies in -> everything published
lead = us(ready - start)            # per slice: how long before its dependencies the wavefront was resident
q = lambda a: [float(np.percentile(a, p)) for p in (5, 25, 50, 75, 95)]
print(json.dumps({"what": "trsv block-kernel trace (shell-like factor)", "slices": int(len(lev)), "levels": nlev,
                  "slices_per_level_q": q(cnt), "total_us": float(us(done.max() - t0)),
                  "per_level_us": float(us(done.max() - t0)) / nlev,
                  "hop_us_q (level l all ready - level l-1 all done)": q(hop),
                  "work_us_q (slice: ready -> done)": q(work),
                  # (the row-by-row shapes do not stamp the phases: null)
                  "lds_us_q (ready -> values in registers)": q(us(t[:, 4] - ready)) if t[:, 4].min() > 0 else None,
                  "ext_us_q (-> external FMAs done)": q(us(t[:, 5] - t[:, 4])) if t[:, 4].min() > 0 else None,
                  "int_us_q (-> rows chained + published)": q(us(done - t[:, 5])) if t[:, 5].min() > 0 else None,
                  "ready_spread_us_q (level: last ready - first ready)": q(us(rd - rdmin)),
                  "lead_us_q (slice: ticket -> ready)": q(lead)}))
