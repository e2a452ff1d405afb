"""Host-side fuzzers of the C ABI that need no GPU: handle creation on malformed arrays, the clean-CSR pass of aoclsparse_optimize
on hostile-but-valid matrices, and the parser of a shipped csrmm state (aoclsparse_mi355_mm_state_adopt).  They run in the plain
CPU tier and -- the reason they exist -- under tests/run_san.sh, where the library's host translation units are built with
-fsanitize=address,undefined (the reference's own sanitizer tier: CMakeLists.txt:118-155)."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as entry  # noqa: E402
import oracle  # noqa: E402
from util import random_csr  # noqa: E402

P = entry.load_package()
L = P.lib()


def _mutate(rng, m, n, base, rp, ci):
    """one random defect (or none) in a valid CSR structure"""
    rp, ci = rp.copy(), ci.copy()
    kind = int(rng.integers(0, 9))
    nnz = len(ci)
    if kind == 1 and nnz:
        ci[rng.integers(0, nnz)] = n + base + int(rng.integers(0, 5))  # column past the end
    elif kind == 2 and nnz:
        ci[rng.integers(0, nnz)] = base - 1 - int(rng.integers(0, 3))  # column before the base
    elif kind == 3 and m > 1:
        i = int(rng.integers(1, m))
        rp[i] = rp[i - 1] - 1 - int(rng.integers(0, 3))  # decreasing row pointer
    elif kind == 4:
        rp[0] = base + 1  # first pointer off the base
    elif kind == 5:
        rp[m] = rp[m] + int(rng.integers(1, 4))  # last pointer past nnz
    elif kind == 6 and nnz > 1:
        j = int(rng.integers(1, nnz))
        ci[j] = ci[j - 1]  # duplicate (maybe across a row boundary: then legal)
    elif kind == 7 and m > 2:
        i = int(rng.integers(1, m))
        rp[i] = rp[m] + 7  # a pointer far past the arrays
    elif kind == 8 and m > 0:
        rp[int(rng.integers(0, m + 1))] = -5
    return rp, ci


def test_create_csr_fuzz_matches_the_oracle_check():
    """aoclsparse_create_dcsr on randomly damaged structures: the status is aoclsparse_mat_check_internal's (oracle.mat_check,
    create/aoclsparse_create.cpp:34-97, analysis/aoclsparse_csr_util.cpp:124-230) and nothing is read out of bounds."""
    rng = np.random.default_rng(2025)
    seen = set()
    for it in range(400):
        m, n, base = int(rng.integers(1, 40)), int(rng.integers(1, 40)), int(rng.integers(0, 2))
        rp, ci, v = random_csr(1000 + it, m, n, lambda r, i: r.integers(0, min(n, 6) + 1), base=base, sort=bool(it & 1))
        rp, ci = _mutate(rng, m, n, base, rp, ci)
        nnz = len(ci)
        h = ctypes.c_void_p()
        vv = v if nnz else np.ones(1)
        st = L.aoclsparse_create_dcsr(ctypes.byref(h), base, m, n, nnz, P._ptr(rp), P._ptr(ci), P._ptr(vv))
        so, _, _ = oracle.mat_check(m, n, nnz, rp, ci, vv, 0, base)
        assert st == so, (it, m, n, base, rp.tolist(), ci.tolist(), st, so)
        seen.add(st)
        if st == 0:
            assert L.aoclsparse_destroy(ctypes.byref(h)) == 0
        else:
            assert not h.value
    assert 0 in seen and len(seen) >= 2


def test_optimize_fuzz_clean_csr_matches_the_oracle():
    """set_sv_hint + aoclsparse_optimize on valid matrices with every nuisance the clean-CSR pass handles -- unsorted rows, missing
    diagonals, empty rows, rectangular shapes, both bases: arrays, idiag and iurow are the oracle's, integer-exact
    (analysis/aoclsparse_csr_util.hpp:765-967)."""
    for it in range(60):
        rng = np.random.default_rng(77 + it)
        m, n, base = int(rng.integers(1, 90)), int(rng.integers(1, 90)), it & 1
        rp, ci, v = random_csr(500 + it, m, n, lambda r, i: 0 if r.random() < 0.15 else r.integers(0, min(n, 9) + 1), base=base,
                               sort=(it % 3 == 0))
        A = P.Matrix(base, m, n, rp, ci, v)
        if A.status != 0:
            continue
        d = P.Descr(base=base, mtype=P.TYPE_TRIANGULAR)
        assert L.aoclsparse_set_sv_hint(A.h, P.OP_NONE, d.h, 1) == 0
        st = L.aoclsparse_optimize(A.h)
        o = oracle.dcsr_optimize(m, n, len(v), base, rp, ci, v)
        assert st == o["status"], (it, st, o["status"])
        if st != 0:
            continue
        e, g = A.export(), A.export_diag()
        assert np.array_equal(e["row_ptr"], o["ptr"]) and np.array_equal(e["col_ind"], o["ind"]) and np.array_equal(e["val"], o["val"])
        assert np.array_equal(g["idiag"], o["idiag"]) and np.array_equal(g["iurow"], o["iurow"])


def test_mm_state_adopt_rejects_mutated_states():
    """a shipped csrmm state is bytes from another process: aoclsparse_mi355_mm_state_adopt must reject every inconsistent header
    before it sizes an allocation or a copy from it, and leave *R NULL"""
    rng = np.random.default_rng(5)
    bufs = (ctypes.c_void_p * P.MM_STATE_BUFFERS)()
    for it in range(300):
        s = P.MmState()
        mode = it % 4
        for i in range(40):
            s.scalars[i] = (0 if mode == 0 else int(rng.integers(-3, 1 << (8 if mode == 1 else 40)))) if mode != 3 else -1
        for i in range(P.MM_STATE_BUFFERS):
            s.bytes[i] = int(rng.integers(-2, 1 << 20)) if mode != 0 else 0
        R = ctypes.c_void_p(0xDEAD)
        st = L.aoclsparse_mi355_mm_state_adopt(ctypes.byref(R), ctypes.byref(s), bufs)
        assert st != 0 and not R.value, (it, st)
    assert L.aoclsparse_mi355_mm_state_adopt(None, None, None) == 2


def test_hint_sequences_and_destroy_never_leak_or_crash():
    """random sequences of the ten hint setters, optimize, set_value / update_values and copy on small handles, then destroy: status
    codes only (analysis/aoclsparse_analysis.cpp:35-385) -- the point is the allocator traffic under AddressSanitizer"""
    rng = np.random.default_rng(9)
    for it in range(40):
        m = int(rng.integers(2, 30))
        rp, ci, v = random_csr(300 + it, m, m, lambda r, i: r.integers(1, min(m, 5) + 1), base=it & 1)
        A = P.Matrix(it & 1, m, m, rp, ci, v)
        assert A.status == 0
        d = P.Descr(base=it & 1)
        for _ in range(int(rng.integers(1, 12))):
            k = int(rng.integers(0, 7))
            if k == 0:
                assert L.aoclsparse_set_mv_hint(A.h, P.OP_NONE, d.h, int(rng.integers(1, 5))) == 0
            elif k == 1:
                assert L.aoclsparse_set_mm_hint(A.h, P.OP_NONE, d.h, 1) == 0
            elif k == 2:
                assert L.aoclsparse_set_2m_hint(A.h, P.OP_NONE, d.h, 1) == 0
            elif k == 3:
                assert L.aoclsparse_set_memory_hint(A.h, int(rng.integers(0, 2))) == 0
            elif k == 4:
                assert L.aoclsparse_optimize(A.h) == 0
            elif k == 5:
                i = int(rng.integers(0, m))
                j = int(ci[rp[i] - (it & 1)]) if rp[i + 1] > rp[i] else 0
                L.aoclsparse_dset_value(A.h, i + (it & 1), j, 2.5)
            else:
                c = ctypes.c_void_p()
                if L.aoclsparse_copy(A.h, d.h, ctypes.byref(c)) == 0:
                    assert L.aoclsparse_destroy(ctypes.byref(c)) == 0
        del A


def test_create_csc_coo_fuzz_accepts_only_valid_structures():
    """aoclsparse_create_dcsc / _dcoo on randomly damaged structures (extra/aoclsparse_auxiliary.cpp: create_csc / create_coo run
    the same structure check as create_csr): a success status implies that every index is inside the matrix and every pointer
    array is monotone and closes at nnz; a rejected input leaves *mat NULL; convert_csr of an accepted COO handle gives a
    structure that create_dcsr accepts."""
    rng = np.random.default_rng(4242)
    outcomes = set()
    for it in range(300):
        m, n, base = int(rng.integers(1, 30)), int(rng.integers(1, 30)), int(rng.integers(0, 2))
        # CSC of an m x n matrix = CSR structure of the n x m transpose
        cp, ri, v = random_csr(7000 + it, n, m, lambda r, i: r.integers(0, min(m, 5) + 1), base=base)
        cp, ri = _mutate(rng, n, m, base, cp, ri)
        nnz = len(ri)
        vv = v if nnz else np.ones(1)
        h = ctypes.c_void_p()
        st = L.aoclsparse_create_dcsc(ctypes.byref(h), base, m, n, nnz, P._ptr(cp), P._ptr(ri), P._ptr(vv))
        ok = (cp[0] == base and cp[n] - base == nnz and np.all(np.diff(cp) >= 0)
              and (nnz == 0 or (ri.min() >= base and ri.max() < m + base)))
        if st == 0:
            assert ok, (it, cp.tolist(), ri.tolist())
            assert L.aoclsparse_destroy(ctypes.byref(h)) == 0
        else:
            assert not h.value
        outcomes.add(st == 0)
        # COO: coordinates with one random defect
        k = int(rng.integers(0, 40))
        rows = rng.integers(0, m, k).astype(np.int32) + base
        cols = rng.integers(0, n, k).astype(np.int32) + base
        if k and rng.random() < 0.4:
            j = int(rng.integers(0, k))
            (rows if rng.random() < 0.5 else cols)[j] = int(rng.choice([-3, base - 1, max(m, n) + base + 2]))
        vals = rng.uniform(-1, 1, max(k, 1))
        rr, cc = (rows, cols) if k else (np.zeros(1, np.int32), np.zeros(1, np.int32))
        h = ctypes.c_void_p()
        st = L.aoclsparse_create_dcoo(ctypes.byref(h), base, m, n, k, P._ptr(rr), P._ptr(cc), P._ptr(vals))
        okc = k == 0 or (rows.min() >= base and rows.max() < m + base and cols.min() >= base and cols.max() < n + base)
        if st == 0:
            assert okc, (it, rows.tolist(), cols.tolist())
            assert L.aoclsparse_destroy(ctypes.byref(h)) == 0
        else:
            assert not h.value
    assert outcomes == {True, False}
