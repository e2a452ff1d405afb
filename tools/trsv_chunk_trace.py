#!/usr/bin/env python3
"""round 6: where does a step of the two-level TRSV schedule (5) spend its time?  One solve of the shell-like ILU(0) factor with
AOCLSPARSE_MI355_TRSV_TRACE set; per step (4 x u64, 100 MHz): taken, first look landed, dependencies in, done.
  python3 tools/trsv_chunk_trace.py [structured|unstructured] [n]"""
import json, os, sys
TRACE = "/tmp/trsv_chunk_trace.bin"
os.environ["AOCLSPARSE_MI355_TRSV_TRACE"] = TRACE
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as entry, oracle, standins
pkg = entry.load_package(); L = pkg.lib()
L.aoclsparse_mi355_set_pointer_mode(pkg.PTR_DEVICE)
assert L.aoclsparse_mi355_set_option(pkg.OPTION_TRSV_CHUNKS, 1) == 0
variant = sys.argv[1] if len(sys.argv) > 1 else "structured"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1508065
m, rp, ci, v = standins.shell_like_unstructured(n=n) if variant == "unstructured" else standins.shell_like(n=n)
st, lu, dg = oracle.dilu0(m, 0, rp, ci, v)
A = pkg.Matrix(0, m, m, rp, ci, lu)
dl = pkg.Descr(mtype=pkg.TYPE_TRIANGULAR, fill=pkg.FILL_LOWER, diag=pkg.DIAG_UNIT)
assert L.aoclsparse_set_sv_hint(A.h, pkg.OP_NONE, dl.h, 100) == 0 and L.aoclsparse_optimize(A.h) == 0
assert L.aoclsparse_mi355_set_trsv_schedule(5) == 0
b = np.random.default_rng(2).uniform(-1, 1, m)
dev = torch.device("cuda", 0)
bd, xd = torch.from_numpy(b).to(dev), torch.zeros(m, dtype=torch.float64, device=dev)
for _ in range(3):
    pkg.dtrsv(pkg.OP_NONE, 1.0, A, dl, bd, xd)
torch.cuda.synchronize()
t = np.fromfile(TRACE, dtype=np.uint64).reshape(-1, 8).astype(np.int64)
ok = t[:, 3] > 0
t = t[ok]
t0 = t[:, 0].min()
if int(os.environ.get("AOCLSPARSE_MI355_TRSV_DBG", "0")) & 8:
    mhz = (t[:, 4] - t[:, 5]) / np.maximum(t[:, 3] - t[:, 0], 1) * 100.0
    print(json.dumps({"shader_clock_MHz_q (s_memtime / s_memrealtime over a step)": [round(float(np.percentile(mhz, p)), 1) for p in (5, 25, 50, 75, 95)]}))
us = lambda x: x / 100.0
q = lambda a: [round(float(np.percentile(a, p)), 3) for p in (5, 25, 50, 75, 95)]
print(json.dumps({"variant": variant, "m": m, "steps": int(len(t)), "total_us": float(us(t[:, 3].max() - t0)),
                  "taken_to_first_look_us_q": q(us(t[:, 1] - t[:, 0])), "first_look_to_ready_us_q": q(us(t[:, 2] - t[:, 1])),
                  "ready_to_done_us_q": q(us(t[:, 3] - t[:, 2])), "step_us_q": q(us(t[:, 3] - t[:, 0])),
                  "ready_to_ext_us_q": q(us(t[:, 4] - t[:, 2])), "ext_to_elim_us_q": q(us(t[:, 5] - t[:, 4])), "elim_to_done_us_q": q(us(t[:, 3] - t[:, 5])),
                  "seen_to_ready_us_q (dbg 4)": q(us(t[:, 2] - t[:, 1])),
                  "hop_us_q (same chunk: ready of step s+1 - elim of step s)": q(us(np.concatenate([
                      (lambda a: a[1:, 2] - a[:-1, 5])(t[t[:, 6] == c][np.argsort(t[t[:, 6] == c][:, 7])]) for c in np.unique(t[:, 6])[:40]]))),
                  "first_steps_done_us": [round(float(us(x - t0)), 2) for x in np.sort(t[:, 3])[:12]],
                  "done_gap_us_q (sorted completion times)": q(us(np.diff(np.sort(t[:, 3]))))}))

for c in (0, 1, int(t[:, 6].max()) // 2):
    a = t[t[:, 6] == c]
    a = a[np.argsort(a[:, 7])][:16]
    print("chunk", c, "steps 0..15: [taken, first look, ready, ext, elim, done] us")
    for r in a:
        print("  step %3d  " % r[7] + "  ".join("%8.2f" % us(r[k] - t0) for k in (0, 1, 2, 4, 5, 3)))
print("chunk: steps, first ready, median ready, last done (us)")
for c in np.unique(t[:, 6]):
    a = t[t[:, 6] == c]
    print("  %3d %5d %9.1f %9.1f %9.1f" % (c, len(a), us(a[:, 2].min() - t0), us(np.median(a[:, 2]) - t0), us(a[:, 3].max() - t0)))
