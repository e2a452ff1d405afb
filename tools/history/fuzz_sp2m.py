#!/usr/bin/env python3
"""round 4: differential fuzz of aoclsparse_sp2m (spgemm_hash_kernel and the device-side analysis) against the CPU oracle.

Random shapes (1 .. 3000 rows, rectangular), index bases 0 / 1 on either operand, op in {N, T} on either side (the oracle's
operand after op = its csr2csc transpose, as the reference's driver builds it), rows that are empty / short / hundreds of entries
long (every LDS bin; with --heavy a few rows whose lists go to the global slab), unsorted rows and repeated off-diagonal columns,
one-stage and two-stage requests.  row_ptr, col_ind and val must be bit-identical.  Prints one JSON line.
  python3 tools/fuzz_sp2m.py [cases=300] [seed=1] [--heavy]"""
import ctypes, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # (tools/history/ -> repository root)
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
import oracle
P = entry.load_package(); L = P.lib()
args = [a for a in sys.argv[1:] if not a.startswith("--")]
cases = int(args[0]) if len(args) > 0 else 300
seed = int(args[1]) if len(args) > 1 else 1
heavy = "--heavy" in sys.argv
rng = np.random.default_rng(seed)


def rand_csr(nr, nc, base, kind):
    """kind 0: short rows; 1: mixed with rows of up to ~300; 2: as 1 plus unsorted rows and repeated columns"""
    lens = rng.integers(0, 7, nr)
    if kind >= 1:
        big = rng.random(nr) < 0.15
        lens = np.where(big, rng.integers(min(20, nc), min(300, nc) + 1, nr), lens)
    if heavy and nr > 40 and nc > 2500:
        lens[rng.integers(0, nr, 2)] = min(nc, int(rng.integers(2200, 3000)))
    lens = np.minimum(lens, nc)
    ptr = np.zeros(nr + 1, np.int64); np.cumsum(lens, out=ptr[1:])
    ind = np.empty(ptr[-1], np.int64)
    for i in range(nr):
        k = lens[i]
        if k == 0:
            continue
        c = np.sort(rng.choice(nc, size=k, replace=False))
        if kind == 2 and k > 2:
            r = rng.random()
            if r < 0.25:
                c = rng.permutation(c)
            elif r < 0.45:
                j = int(rng.integers(1, k))
                if c[0] != i:
                    c[j] = c[0]  # a repeated off-diagonal column (a repeated DIAGONAL is refused at creation)
        ind[ptr[i]:ptr[i + 1]] = c
    val = rng.uniform(-1, 1, ptr[-1])
    return (ptr + base).astype(np.int32), (ind + base).astype(np.int32), val


def export(h):
    base, m, n, nnz = ctypes.c_int(), ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
    rp, ci, v = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
    assert L.aoclsparse_export_dcsr(h, ctypes.byref(base), ctypes.byref(m), ctypes.byref(n), ctypes.byref(nnz), ctypes.byref(rp),
                                    ctypes.byref(ci), ctypes.byref(v)) == 0
    k = max(nnz.value, 1)
    row = np.ctypeslib.as_array(ctypes.cast(rp, ctypes.POINTER(ctypes.c_int32)), (m.value + 1,)).copy()
    col = np.ctypeslib.as_array(ctypes.cast(ci, ctypes.POINTER(ctypes.c_int32)), (k,))[: nnz.value].copy()
    val = np.ctypeslib.as_array(ctypes.cast(v, ctypes.POINTER(ctypes.c_double)), (k,))[: nnz.value].copy()
    return base.value, m.value, n.value, row, col, val


def operand(m, n, base, p, i, v, trans):
    if not trans:
        return m, n, p, i, v
    st, cp, ri, cv = oracle.dcsr2csc(m, n, int(p[m] - base), base, base, p, i, v)
    assert st == 0
    return n, m, cp, ri.astype(np.int32), cv


t0 = time.time()
bad = []
stats = {"cases": 0, "two_stage": 0, "transposed": 0, "max_row_of_c": 0, "nnz_c_total": 0}
for case in range(cases):
    big = heavy or rng.random() < 0.3
    inner = int(rng.integers(1, 3000 if big else 200))
    rows_c = int(rng.integers(1, 3000 if big else 200))
    cols_c = int(rng.integers(1, 4000 if big else 200))
    ta, tb = bool(rng.random() < 0.3), bool(rng.random() < 0.3)
    ba, bb = int(rng.integers(0, 2)), int(rng.integers(0, 2))
    kind = int(rng.integers(0, 3))
    # stored shapes: op(A) is rows_c x inner, op(B) is inner x cols_c
    am, an = (inner, rows_c) if ta else (rows_c, inner)
    bm, bn = (cols_c, inner) if tb else (inner, cols_c)
    if ta and tb:
        pass  # (B A)^T path
    pa, ia, va = rand_csr(am, an, ba, kind)
    pb, ib, vb = rand_csr(bm, bn, bb, kind)
    A, B = P.Matrix(ba, am, an, pa, ia, va), P.Matrix(bb, bm, bn, pb, ib, vb)
    if A.status != 0 or B.status != 0:
        bad.append({"case": case, "what": "create", "status": [A.status, B.status]}); continue
    dA, dB = P.Descr(base=ba), P.Descr(base=bb)
    xm, xn, xp, xi, xv = operand(am, an, ba, pa, ia, va, ta)
    ym, yn, yp, yi, yv = operand(bm, bn, bb, pb, ib, vb, tb)
    so, pc, ic, vc = oracle.dcsr2m(xm, yn, ba, xp, xi, xv, bb, yp, yi, yv)
    assert so == 0
    if ta and tb:
        # the reference forms (B A)^T: D = B * A (stored operands), then C = D^T by its counting-sort transpose
        so, pd, idd, vd = oracle.dcsr2m(bm, an, bb, pb, ib, vb, ba, pa, ia, va)
        st, cp, ri, cv = oracle.dcsr2csc(bm, an, len(idd), 0, 0, pd, idd, vd)
        pc, ic, vc = cp, ri.astype(np.int32), cv
    opa = P.OP_TRANSPOSE if ta else P.OP_NONE
    opb = P.OP_TRANSPOSE if tb else P.OP_NONE
    C = ctypes.c_void_p()
    two = bool(rng.random() < 0.4)
    if two:
        st = L.aoclsparse_sp2m(opa, dA.h, A.h, opb, dB.h, B.h, P.STAGE_NNZ_COUNT, ctypes.byref(C))
        if st == 0:
            st = L.aoclsparse_sp2m(opa, dA.h, A.h, opb, dB.h, B.h, P.STAGE_FINALIZE, ctypes.byref(C))
    else:
        st = L.aoclsparse_sp2m(opa, dA.h, A.h, opb, dB.h, B.h, P.STAGE_FULL, ctypes.byref(C))
    if st != 0:
        bad.append({"case": case, "what": "status", "status": st}); continue
    b, cm, cn, row, col, val = export(C)
    ok = b == 0 and (cm, cn) == (rows_c, cols_c) and np.array_equal(row, pc) and np.array_equal(col, ic) and np.array_equal(val, vc)
    if not ok:
        bad.append({"case": case, "what": "mismatch", "shape": [rows_c, inner, cols_c], "ops": [ta, tb], "bases": [ba, bb], "kind": kind,
                    "two_stage": two, "row_ptr_ok": bool(np.array_equal(row, pc)), "col_ok": bool(len(col) == len(ic) and np.array_equal(col, ic))})
    stats["cases"] += 1; stats["two_stage"] += two; stats["transposed"] += ta or tb
    stats["nnz_c_total"] += int(len(ic)); stats["max_row_of_c"] = max(stats["max_row_of_c"], int(np.diff(pc).max()) if len(pc) > 1 else 0)
    L.aoclsparse_destroy(ctypes.byref(C))
print(json.dumps({"tool": "fuzz_sp2m", "seed": seed, "heavy": heavy, **stats, "mismatches": len(bad), "first": bad[:3], "seconds": round(time.time() - t0, 1)}))
sys.exit(1 if bad else 0)
