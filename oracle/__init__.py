"""ctypes binding of the CPU parity oracle (oracle/oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package never imports this module.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

c_int = ctypes.c_int
c_i32 = ctypes.c_int32
c_dbl = ctypes.c_double
c_flt = ctypes.c_float
P = ctypes.c_void_p


def build(force=False):
    """Compile oracle.c with the committed Makefile (gcc only)."""
    src = os.path.join(_HERE, "oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(
        os.path.getmtime(src), os.path.getmtime(os.path.join(_HERE, "oracle.h"))
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s", "clean", "all"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
    return _lib


def _p(a):
    return a.ctypes.data_as(P) if a is not None else None


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def dcsrmv(kid, base, alpha, m, nnz, val, col, row, x, beta, y, nthreads=0):
    """y <- alpha*A*x + beta*y with the reference's dispatch rule; returns (status, y)."""
    val, col, row, x = _f64(val), _i32(col), _i32(row), _f64(x)
    y = _f64(y).copy()
    L = lib()
    if nthreads and nthreads > 0:
        st = L.orc_dcsrmv_omp(c_int(kid), c_int(base), c_dbl(alpha), c_i32(m), c_i32(nnz), _p(val),
                              _p(col), _p(row), _p(x), c_dbl(beta), _p(y), c_int(nthreads))
    else:
        st = L.orc_dcsrmv(c_int(kid), c_int(base), c_dbl(alpha), c_i32(m), c_i32(nnz), _p(val),
                          _p(col), _p(row), _p(x), c_dbl(beta), _p(y))
    return st, y


def dcsrmv_inplace(kid, base, alpha, m, nnz, val, col, row, x, beta, y, nthreads=1):
    """Timing leg: arrays must already be contiguous numpy arrays of the right dtype; y is updated in
    place (no copies inside the timed call)."""
    return lib().orc_dcsrmv_omp(c_int(kid), c_int(base), c_dbl(alpha), c_i32(m), c_i32(nnz), _p(val), _p(col),
                                _p(row), _p(x), c_dbl(beta), _p(y), c_int(nthreads))


def dcsrmv_bench(kid, base, m, n, nnz, val, col, row, x, nthreads, passes):
    """CPU-baseline leg: `passes` SpMVs (alpha=1, beta=0) on first-touched copies of the arrays, each pass timed on
    its own -> (status, seconds[passes], y of the last pass)."""
    val, col, row, x = _f64(val), _i32(col), _i32(row), _f64(x)
    secs = np.zeros(max(passes, 1), dtype=np.float64)
    y = np.zeros(max(m, 1), dtype=np.float64)
    st = lib().orc_dcsrmv_bench(c_int(kid), c_int(base), c_i32(m), c_i32(n), c_i32(nnz), _p(val), _p(col), _p(row),
                                _p(x), c_int(nthreads), c_int(passes), _p(secs), _p(y))
    return st, secs[:passes], y[:m]


def dcsrmv_order(order, base, alpha, m, val, col, row, x, beta, y):
    """order in {'ref','lane4','lane8'}: one specific reference kernel."""
    fn = {"ref": "orc_dcsrmv_ref", "lane4": "orc_dcsrmv_lane4", "lane8": "orc_dcsrmv_lane8"}[order]
    val, col, row, x = _f64(val), _i32(col), _i32(row), _f64(x)
    y = _f64(y).copy()
    st = getattr(lib(), fn)(c_int(base), c_dbl(alpha), c_i32(m), _p(val), _p(col), _p(row), _p(x),
                            c_dbl(beta), _p(y))
    return st, y


def scsrmv(order, base, alpha, m, val, col, row, x, beta, y):
    fn = {"ref": "orc_scsrmv_ref", "lane8": "orc_scsrmv_lane8"}[order]
    val, col, row, x = _f32(val), _i32(col), _i32(row), _f32(x)
    y = _f32(y).copy()
    st = getattr(lib(), fn)(c_int(base), c_flt(alpha), c_i32(m), _p(val), _p(col), _p(row), _p(x),
                            c_flt(beta), _p(y))
    return st, y


def dcsrmvt(base, alpha, m, n, val, col, row, x, beta, y):
    val, col, row, x = _f64(val), _i32(col), _i32(row), _f64(x)
    y = _f64(y).copy()
    st = lib().orc_dcsrmvt(c_int(base), c_dbl(alpha), c_i32(m), c_i32(n), _p(val), _p(col), _p(row),
                           _p(x), c_dbl(beta), _p(y))
    return st, y


def dcsrmv_symm_raw(base, alpha, m, val, col, row, x, beta, y):
    val, col, row, x = _f64(val), _i32(col), _i32(row), _f64(x)
    y = _f64(y).copy()
    st = lib().orc_dcsrmv_symm_raw(c_int(base), c_dbl(alpha), c_i32(m), _p(val), _p(col), _p(row), _p(x),
                                   c_dbl(beta), _p(y))
    return st, y


def dcsrmv_special(kind, base, alpha, m, n, diag, fill, val, col, ptr, idiag, iurow, x, beta, y):
    """kind: 'symm' | 'tri' | 'tri_t' on the clean CSR (+ idiag/iurow); fill 0/1, diag 0/1/2."""
    val, col, ptr, idiag, iurow, x = _f64(val), _i32(col), _i32(ptr), _i32(idiag), _i32(iurow), _f64(x)
    y = _f64(y).copy()
    L = lib()
    if kind == "symm":
        st = L.orc_dcsrmv_symm(c_int(base), c_dbl(alpha), c_i32(m), c_int(diag), c_int(fill), _p(val), _p(col),
                               _p(ptr), _p(idiag), _p(iurow), _p(x), c_dbl(beta), _p(y))
    elif kind == "tri":
        st = L.orc_dcsrmv_tri(c_int(base), c_dbl(alpha), c_i32(m), c_int(diag), c_int(fill), _p(val), _p(col),
                              _p(ptr), _p(idiag), _p(iurow), _p(x), c_dbl(beta), _p(y))
    else:
        st = L.orc_dcsrmv_tri_t(c_int(base), c_dbl(alpha), c_i32(m), c_i32(n), c_int(diag), c_int(fill), _p(val),
                                _p(col), _p(ptr), _p(idiag), _p(iurow), _p(x), c_dbl(beta), _p(y))
    return st, y


def max_threads():
    return int(lib().orc_max_threads())


def dtrsv(kind, alpha, m, base, a, icol, ilrow, ilend, b, unit, incb=1, incx=1, x0=None):
    """kind in {'l','lt','u','ut'}; ilend = idiag (l, lt) or iurow (u, ut)."""
    a, icol, ilrow, ilend, b = _f64(a), _i32(icol), _i32(ilrow), _i32(ilend), _f64(b)
    x = np.zeros(max(1, (m - 1) * incx + 1), dtype=np.float64) if x0 is None else _f64(x0).copy()
    fn = getattr(lib(), "orc_dtrsv_" + kind)
    st = fn(c_dbl(alpha), c_i32(m), c_int(base), _p(a), _p(icol), _p(ilrow), _p(ilend), _p(b),
            c_i32(incb), _p(x), c_i32(incx), c_int(1 if unit else 0))
    return st, x


def dcsrmm(order, alpha, base, val, col, row, m, B, n, ldb, beta, C, ldc):
    """order: 'row' or 'col' (the two reference kernels)."""
    val, col, row, B = _f64(val), _i32(col), _i32(row), _f64(B)
    C = _f64(C).copy()
    fn = lib().orc_dcsrmm_row if order == "row" else lib().orc_dcsrmm_col
    st = fn(c_dbl(alpha), c_int(base), _p(val), _p(col), _p(row), c_i32(m), _p(B), c_i32(n),
            c_i32(ldb), c_dbl(beta), _p(C), c_i32(ldc))
    return st, C


def set_contract(fused):
    """True (default): the reference's scalar accumulation loops are fused multiply-adds (clang / AOCC build);
    False: a multiplication and an addition (GCC build with the reference's own -march=znver2: oracle.c header)."""
    lib().orc_set_contract(c_int(1 if fused else 0))


class contract:
    """with oracle.contract(False): ... -- scoped switch, restored on exit."""

    def __init__(self, fused):
        self.fused = fused

    def __enter__(self):
        self.prev = lib().orc_get_contract()
        set_contract(self.fused)

    def __exit__(self, *a):
        set_contract(bool(self.prev))


def kt_hsum(tsz, v):
    """kt_hsum_p of one vector register (double: tsz 4 / 8, float32: 8 / 16) as restated in oracle.c."""
    v = np.ascontiguousarray(v)
    L = lib()
    if v.dtype == np.float32:
        L.orc_kt_hsum_s.restype = c_flt
        return np.float32(L.orc_kt_hsum_s(c_int(tsz), _p(v)))
    v = _f64(v)
    L.orc_kt_hsum_d.restype = c_dbl
    return L.orc_kt_hsum_d(c_int(tsz), _p(v))


def trsv_kt(kind, tsz, alpha, m, base, a, icol, ilrow, ilend, b, unit, incb=1, incx=1, x0=None, dtype=np.float64):
    """The KT kernels kt_trsv_{l,lt,u,ut} (trsv_kt.cpp:64-531); tsz = lanes (double 4: kid 1/2, 8: kid 3; float 8 / 16)."""
    f = _f64 if dtype == np.float64 else _f32
    a, icol, ilrow, ilend, b = f(a), _i32(icol), _i32(ilrow), _i32(ilend), f(b)
    x = np.zeros(max(1, (m - 1) * incx + 1), dtype=dtype) if x0 is None else f(x0).copy()
    fn = getattr(lib(), "orc_%strsv_kt_%s" % ("d" if dtype == np.float64 else "s", kind))
    sc = c_dbl if dtype == np.float64 else c_flt
    st = fn(c_int(tsz), sc(alpha), c_i32(m), c_int(base), _p(a), _p(icol), _p(ilrow), _p(ilend), _p(b),
            c_i32(incb), _p(x), c_i32(incx), c_int(1 if unit else 0))
    return st, x


def dcsrmm_kt(order, psz, alpha, base, val, col, row, m, B, n, ldb, beta, C, ldc):
    """csrmm_col_kt / csrmm_row_kt (csrmm_kt.cpp:31-363); psz = 4 (kid 1/2) or 8 (kid 3)."""
    val, col, row, B = _f64(val), _i32(col), _i32(row), _f64(B)
    C = _f64(C).copy()
    fn = lib().orc_dcsrmm_row_kt if order == "row" else lib().orc_dcsrmm_col_kt
    st = fn(c_int(psz), c_dbl(alpha), c_int(base), _p(val), _p(col), _p(row), c_i32(m), _p(B), c_i32(n),
            c_i32(ldb), c_dbl(beta), _p(C), c_i32(ldc))
    return st, C


def scsrmm_kt(order, psz, alpha, base, val, col, row, m, B, n, ldb, beta, C, ldc):
    """float csrmm_col_kt / csrmm_row_kt; psz = 8 (kid 1/2) or 16 (kid 3)."""
    val, col, row, B = _f32(val), _i32(col), _i32(row), _f32(B)
    C = _f32(C).copy()
    fn = lib().orc_scsrmm_row_kt if order == "row" else lib().orc_scsrmm_col_kt
    st = fn(c_int(psz), c_flt(alpha), c_int(base), _p(val), _p(col), _p(row), c_i32(m), _p(B), c_i32(n),
            c_i32(ldb), c_flt(beta), _p(C), c_i32(ldc))
    return st, C


_ktref = None


def ktref():
    """oracle/_ref/libktref.so: the reference's own kernel-template micro-kernels behind ref_kt_driver.cpp (built by
    `make -C oracle ktref` where /root/reference exists).  None when absent or when the host lacks AVX-512."""
    global _ktref
    if _ktref is None:
        path = os.path.join(_HERE, "_ref", "libktref.so")
        if not os.path.exists(path):
            return None
        try:
            with open("/proc/cpuinfo") as f:
                flags = f.read()
        except OSError:
            flags = ""
        if not all(k in flags for k in ("avx512f", "avx512vl", "avx512dq")):
            return None
        L = ctypes.CDLL(path)
        for name in ("ktref_hsum_d", "ktref_dot_d", "ktref_trsv_row_d", "ktref_csrmm_col_elem_d"):
            getattr(L, name).restype = c_dbl
        for name in ("ktref_hsum_s", "ktref_dot_s", "ktref_trsv_row_s", "ktref_csrmm_col_elem_s"):
            getattr(L, name).restype = c_flt
        _ktref = L
    return _ktref


def dscale_dense(order, C, m, n, ld, beta):
    C = _f64(C).copy()
    st = lib().orc_dscale_dense(c_int(1 if order == "col" else 0), _p(C), c_i32(m), c_i32(n),
                                c_i32(ld), c_dbl(beta))
    return st, C


def mat_check(maj, mind, nnz, ptr, ind, val, shape, base):
    ptr, ind, val = _i32(ptr), _i32(ind), _f64(val)
    sort, fd = c_int(0), c_int(0)
    st = lib().orc_mat_check(c_i32(maj), c_i32(mind), c_i32(nnz), _p(ptr), _p(ind), _p(val),
                             c_int(shape), c_int(base), ctypes.byref(sort), ctypes.byref(fd))
    return st, sort.value, bool(fd.value)


def csr_indices(m, base, ptr, ind):
    ptr, ind = _i32(ptr), _i32(ind)
    idiag = np.zeros(max(m, 1), dtype=np.int32)
    iurow = np.zeros(max(m, 1), dtype=np.int32)
    st = lib().orc_csr_indices(c_i32(m), c_int(base), _p(ptr), _p(ind), _p(idiag), _p(iurow))
    return st, idiag[:m], iurow[:m]


def dcsr_optimize(m, n, nnz, base, ptr, ind, val):
    """Returns dict(status, ptr, ind, val, idiag, iurow, is_internal, fulldiag, base)."""
    ptr, ind, val = _i32(ptr), _i32(ind), _f64(val)
    cap = nnz + min(m, n) + 1
    optr = np.zeros(m + 1, dtype=np.int32)
    oind = np.zeros(cap, dtype=np.int32)
    oval = np.zeros(cap, dtype=np.float64)
    idiag = np.zeros(max(m, 1), dtype=np.int32)
    iurow = np.zeros(max(m, 1), dtype=np.int32)
    onnz, internal, fd = c_i32(0), c_int(0), c_int(0)
    st = lib().orc_dcsr_optimize(c_i32(m), c_i32(n), c_i32(nnz), c_int(base), _p(ptr), _p(ind),
                                 _p(val), _p(optr), _p(oind), _p(oval), ctypes.byref(onnz),
                                 _p(idiag), _p(iurow), ctypes.byref(internal), ctypes.byref(fd))
    if st == 0 and not internal.value:
        optr, oind, oval, obase = ptr, ind[:nnz], val[:nnz], base
    else:
        oind, oval, obase = oind[: onnz.value], oval[: onnz.value], 0
    return dict(status=st, ptr=optr, ind=oind, val=oval, idiag=idiag[:m], iurow=iurow[:m],
                is_internal=bool(internal.value), fulldiag=bool(fd.value), base=obase)


def dcsr2csc(m, n, nnz, base_csr, base_csc, row_ptr, col_ind, val):
    row_ptr, col_ind, val = _i32(row_ptr), _i32(col_ind), _f64(val)
    ri = np.zeros(max(nnz, 1), dtype=np.int32)
    cp = np.zeros(n + 1, dtype=np.int32)
    cv = np.zeros(max(nnz, 1), dtype=np.float64)
    st = lib().orc_dcsr2csc(c_i32(m), c_i32(n), c_i32(nnz), c_int(base_csr), c_int(base_csc),
                            _p(row_ptr), _p(col_ind), _p(val), _p(ri), _p(cp), _p(cv))
    return st, cp, ri[:nnz], cv[:nnz]


def dilu0(n, base, row_ptr, col_ind, val):
    row_ptr, col_ind = _i32(row_ptr), _i32(col_ind)
    val = _f64(val).copy()
    diag = np.zeros(max(n, 1), dtype=np.int32)
    st = lib().orc_dilu0(c_i32(n), c_int(base), _p(diag), _p(val), _p(row_ptr), _p(col_ind))
    return st, val, diag[:n]


def dilu_solve(n, base, lu_diag_ptr, val, row_ptr, col_ind, b):
    """x = U^-1 L^-1 b with the ILU(0) factors (values `val` on the pattern row_ptr/col_ind)."""
    val, row_ptr, col_ind, b, lu = _f64(val), _i32(row_ptr), _i32(col_ind), _f64(b), _i32(lu_diag_ptr)
    x = np.zeros(n, dtype=np.float64)
    st = lib().orc_dilu_solve(c_i32(n), c_int(base), _p(lu), _p(val), _p(row_ptr), _p(col_ind), _p(x), _p(b))
    return st, x


def dsymgs(mtype, fill, trans, base, alpha, m, val, col, ptr, idiag, iurow, b, x0):
    """One symmetric Gauss-Seidel sweep on the clean CSR; mtype 0 general / 1 symmetric / 3 triangular."""
    val, col, ptr, idiag, iurow, b = _f64(val), _i32(col), _i32(ptr), _i32(idiag), _i32(iurow), _f64(b)
    x = _f64(x0).copy()
    st = lib().orc_dsymgs(c_int(mtype), c_int(fill), c_int(trans), c_int(base), c_dbl(alpha), c_i32(m), _p(val),
                          _p(col), _p(ptr), _p(idiag), _p(iurow), _p(b), _p(x), None, c_int(0))
    return st, x


def csr2ell(layout, m, base, row_ptr, col_ind, val):
    """layout 'ell' | 'ellt' -> (width, ell_col, ell_val); 'hyb' -> (width, ell_m, map, ell_col, ell_val)."""
    row_ptr, col_ind, val = _i32(row_ptr), _i32(col_ind), _f64(val)
    L = lib()
    w, em = c_i32(0), c_i32(0)
    if layout == "hyb":
        L.orc_csr2ellthyb_width(c_i32(m), c_i32(len(val)), _p(row_ptr), ctypes.byref(em), ctypes.byref(w))
    else:
        L.orc_csr2ell_width(c_i32(m), _p(row_ptr), ctypes.byref(w))
    cells = max(1, m * w.value)
    ec, ev = np.zeros(cells, np.int32), np.zeros(cells, np.float64)
    if layout == "hyb":
        mp = np.zeros(max(1, m - em.value), np.int32)
        em2 = c_i32(0)
        L.orc_dcsr2ellthyb(c_i32(m), c_int(base), ctypes.byref(em2), _p(row_ptr), _p(col_ind), _p(val), _p(mp), _p(ec),
                           _p(ev), w)
        assert em2.value == em.value
        return w.value, em.value, mp[: m - em.value], ec[: m * w.value], ev[: m * w.value]
    L.orc_dcsr2ell(c_int(1 if layout == "ellt" else 0), c_i32(m), c_int(base), _p(row_ptr), _p(col_ind), _p(val), _p(ec),
                   _p(ev), w)
    return w.value, ec[: m * w.value], ev[: m * w.value]


def dellmv(layout, base, alpha, m, val, col, width, x, beta, y):
    val, col, x = _f64(val), _i32(col), _f64(x)
    y = _f64(y).copy()
    fn = lib().orc_delltmv if layout == "ellt" else lib().orc_dellmv
    st = fn(c_int(base), c_dbl(alpha), c_i32(m), _p(val), _p(col), c_i32(width), _p(x), c_dbl(beta), _p(y))
    return st, y


def sellmv(base, alpha, m, val, col, width, x, beta, y):
    val, col, x = _f32(val), _i32(col), _f32(x)
    y = _f32(y).copy()
    st = lib().orc_sellmv(c_int(base), ctypes.c_float(alpha), c_i32(m), _p(val), _p(col), c_i32(width), _p(x),
                          ctypes.c_float(beta), _p(y))
    return st, y


def dellthybmv(base, alpha, m, ell_val, ell_col, width, ell_m, csr_val, csr_row, csr_col, rmap, x, beta, y):
    ell_val, ell_col, csr_val, csr_row, csr_col = _f64(ell_val), _i32(ell_col), _f64(csr_val), _i32(csr_row), _i32(csr_col)
    rmap, x = _i32(rmap), _f64(x)
    y = _f64(y).copy()
    st = lib().orc_dellthybmv(c_int(base), c_dbl(alpha), c_i32(m), _p(ell_val), _p(ell_col), c_i32(width), c_i32(ell_m),
                              _p(csr_val), _p(csr_row), _p(csr_col), _p(rmap), _p(x), c_dbl(beta), _p(y))
    return st, y


def dcsrsv(lower, unit, alpha, m, val, col, row_ptr, x):
    val, col, row_ptr, x = _f64(val), _i32(col), _i32(row_ptr), _f64(x)
    y = np.zeros(m)
    st = lib().orc_dcsrsv(c_int(1 if lower else 0), c_int(1 if unit else 0), c_dbl(alpha), c_i32(m), _p(val), _p(col),
                          _p(row_ptr), _p(x), _p(y))
    return st, y


def opt_blksize(m, nnz, base, row_ptr, col_ind):
    """-> (rows_blk or 0, total blocks)"""
    row_ptr, col_ind = _i32(row_ptr), _i32(col_ind)
    tot = c_i32(0)
    fn = lib().orc_opt_blksize
    fn.restype = c_i32
    r = fn(c_i32(m), c_i32(nnz), c_int(base), _p(row_ptr), _p(col_ind), ctypes.byref(tot))
    return r, tot.value


def csr2blkcsr(m, n, base, row_ptr, col_ind, val, rows_blk):
    """-> (status, blk_row_ptr, blk_col_ind, blk_val, masks) trimmed to the blocks produced"""
    row_ptr, col_ind, val = _i32(row_ptr), _i32(col_ind), _f64(val)
    nnz = len(val)
    brp = np.zeros(m + 1, np.int32)
    bc = np.zeros(max(1, nnz), np.int32)
    bv = np.zeros(nnz + rows_blk * 8, np.float64)
    mk = np.zeros(max(1, nnz) * rows_blk + rows_blk * 8, np.uint8)
    nb = c_i32(0)
    st = lib().orc_dcsr2blkcsr(c_i32(m), c_i32(n), c_i32(nnz), _p(row_ptr), _p(col_ind), _p(val), _p(brp), _p(bc), _p(bv),
                               _p(mk), c_i32(rows_blk), c_int(base), ctypes.byref(nb))
    return st, brp, bc[: nb.value], bv[:nnz], mk[: nb.value * rows_blk]


def dblkcsrmv(base, alpha, m, masks, blk_val, blk_col, blk_row_ptr, x, beta, y, rows_blk):
    masks = np.ascontiguousarray(masks, dtype=np.uint8)
    blk_val, blk_col, blk_row_ptr, x = _f64(blk_val), _i32(blk_col), _i32(blk_row_ptr), _f64(x)
    y = _f64(y).copy()
    st = lib().orc_dblkcsrmv(c_int(base), c_dbl(alpha), c_i32(m), _p(masks), _p(blk_val), _p(blk_col), _p(blk_row_ptr),
                             _p(x), c_dbl(beta), _p(y), c_i32(rows_blk))
    return st, y


def dcg(n, base, ptr, col, val, idiag, iurow, b, x0, rtol, atol, maxit, precond):
    """CG on a clean CSR holding the whole symmetric matrix; precond 0 none / 3 SymGS -> (status, x, rinfo)."""
    ptr, col, val, idiag, iurow, b = _i32(ptr), _i32(col), _f64(val), _i32(idiag), _i32(iurow), _f64(b)
    x, rinfo = _f64(x0).copy(), np.zeros(100)
    st = lib().orc_dcg(c_i32(n), c_int(base), _p(ptr), _p(col), _p(val), _p(idiag), _p(iurow), _p(b), _p(x),
                       c_dbl(rtol), c_dbl(atol), c_i32(maxit), c_int(precond), _p(rinfo))
    return st, x, rinfo


def dgmres(n, base, ptr, col, val, b, x0, restart, rtol, atol, maxit, precond):
    """restarted GMRES; precond 0 none / 2 ILU0 -> (status, x, rinfo)."""
    ptr, col, val, b = _i32(ptr), _i32(col), _f64(val), _f64(b)
    x, rinfo = _f64(x0).copy(), np.zeros(100)
    st = lib().orc_dgmres(c_i32(n), c_int(base), _p(ptr), _p(col), _p(val), _p(b), _p(x), c_i32(restart), c_dbl(rtol),
                          c_dbl(atol), c_i32(maxit), c_int(precond), _p(rinfo))
    return st, x, rinfo


def zmv(op, mtype, fill, diag, base, alpha, m, n, ptr, ind, val, x, beta, y):
    """Complex y = alpha op(M) x + beta y, M assembled from a CLEAN CSR (sorted, unique) the way the reference's
    doid dispatch reads it (level2/aoclsparse_mv.cpp:41-349, csrmv_kr.hpp:41-444): mtype 'general' | 'symmetric'
    | 'hermitian' | 'triangular'; fill 'lower' | 'upper'; diag 'non_unit' | 'unit' | 'zero'; op 'n' | 't' | 'h'.
    numpy restatement (complex128 accumulate; also used for complex64 inputs with a float32 tolerance)."""
    ptr, ind = np.asarray(ptr, np.int64) - base, np.asarray(ind, np.int64) - base
    val, x, y = np.asarray(val, np.complex128), np.asarray(x, np.complex128), np.asarray(y, np.complex128)
    rows = np.repeat(np.arange(m), np.diff(ptr))
    M = np.zeros((m, n), np.complex128)
    if mtype == "general":
        np.add.at(M, (rows, ind), val)
    else:
        strict = (ind < rows) if fill == "lower" else (ind > rows)
        np.add.at(M, (rows[strict], ind[strict]), val[strict])
        if mtype == "symmetric":
            M = M + M.T
        elif mtype == "hermitian":
            M = M + M.conj().T
        k = min(m, n)
        if diag == "unit":
            M[np.arange(k), np.arange(k)] += 1.0
        elif diag == "non_unit":
            d = ind == rows
            np.add.at(M, (rows[d], ind[d]), val[d])
    Mo = {"n": M, "t": M.T, "h": M.conj().T}[op]
    scale = np.abs(alpha) * (np.abs(Mo) @ np.abs(x)) + np.abs(beta) * np.abs(y)
    return alpha * (Mo @ x) + (beta * y if beta != 0 else 0.0), scale


def dcsr2m(m, n, base_a, ptr_a, ind_a, val_a, base_b, ptr_b, ind_b, val_b):
    """C = A*B (general CSR x CSR); C is 0-based, columns in first-touch order."""
    ptr_a, ind_a, val_a = _i32(ptr_a), _i32(ind_a), _f64(val_a)
    ptr_b, ind_b, val_b = _i32(ptr_b), _i32(ind_b), _f64(val_b)
    ptr_c = np.zeros(m + 1, dtype=np.int32)
    nnz_c = c_i32(0)
    st = lib().orc_csr2m_nnz(c_i32(m), c_i32(n), c_int(base_a), _p(ptr_a), _p(ind_a), c_int(base_b),
                             _p(ptr_b), _p(ind_b), _p(ptr_c), ctypes.byref(nnz_c))
    if st != 0:
        return st, None, None, None
    ind_c = np.zeros(max(nnz_c.value, 1), dtype=np.int32)
    val_c = np.zeros(max(nnz_c.value, 1), dtype=np.float64)
    st = lib().orc_dcsr2m_fill(c_i32(m), c_i32(n), c_int(base_a), _p(ptr_a), _p(ind_a), _p(val_a),
                               c_int(base_b), _p(ptr_b), _p(ind_b), _p(val_b), _p(ptr_c), _p(ind_c),
                               _p(val_c))
    return st, ptr_c, ind_c[: nnz_c.value], val_c[: nnz_c.value]


def _op_operand(m, n, base, ptr, ind, val, trans):
    """The operand after op handling: the stable csr2csc transpose the reference's drivers build (same base)."""
    ptr, ind, val = _i32(ptr), _i32(ind), _f64(val)
    if not trans:
        return m, n, ptr, ind, val
    nnz = int(ptr[m] - base)
    st, cp, ri, cv = dcsr2csc(m, n, nnz, base, base, ptr, ind, val)
    assert st == 0
    return n, m, cp, _i32(ri), _f64(cv)


def dsp2md(a, trans_a, b, trans_b, alpha, beta, C, rowmajor, ldc):
    """C = alpha*op(A)*op(B) + beta*C, dense C (level3/aoclsparse_sp2md.hpp:179-431).  a / b = (m, n, base, ptr, ind,
    val); C is the flat outer x ldc storage.  Returns the updated copy."""
    ma, na, pa, ia, va = _op_operand(*a, trans_a)
    mb, nb, pb, ib, vb = _op_operand(*b, trans_b)
    assert na == mb
    C = np.array(C, dtype=np.float64).ravel().copy()
    outer, inner = (ma, nb) if rowmajor else (nb, ma)
    lib().orc_dsp2md_scale(c_i32(outer), c_i32(inner), c_i32(ldc), c_dbl(beta), _p(C))
    if alpha != 0.0:
        rs, cs = (ldc, 1) if rowmajor else (1, ldc)
        lib().orc_dsp2md(c_i32(ma), c_int(a[2]), _p(pa), _p(ia), _p(va), c_int(b[2]), _p(pb), _p(ib), _p(vb),
                         c_dbl(alpha), _p(C), ctypes.c_longlong(rs), ctypes.c_longlong(cs))
    return C


def dcsr2dense(m, n, base, ptr, ind, val, A, ld, colmajor, mode=0, fill=0, diag=0):
    """conversion/aoclsparse_convert.hpp:658-929; A is the flat outer x ld storage (padding kept)."""
    ptr, ind, val = _i32(ptr), _i32(ind), _f64(val)
    A = np.array(A, dtype=np.float64).ravel().copy()
    lib().orc_dcsr2dense(c_i32(m), c_i32(n), c_int(base), _p(ptr), _p(ind), _p(val), _p(A), c_i32(ld),
                         c_int(1 if colmajor else 0), c_int(mode), c_int(fill), c_int(diag))
    return A


def dcsradd(a, trans_a, alpha, b):
    """C = alpha*op(A) + B (level3/aoclsparse_csradd.hpp:283-532).  Returns (ptr, ind, val) in A's base."""
    ma, na, pa, ia, va = _op_operand(*a, trans_a)
    mb, nb, base_b, pb, ib, vb = b[0], b[1], b[2], _i32(b[3]), _i32(b[4]), _f64(b[5])
    assert (ma, na) == (mb, nb)
    cap = max(len(ia) + len(ib), 1)
    pc = np.zeros(ma + 1, dtype=np.int32)
    ic = np.zeros(cap, dtype=np.int32)
    vc = np.zeros(cap, dtype=np.float64)
    lib().orc_dcsradd.restype = c_i32
    w = lib().orc_dcsradd(c_i32(ma), c_i32(na), c_int(a[2]), _p(pa), _p(ia), _p(va), c_dbl(alpha), c_int(base_b),
                          _p(pb), _p(ib), _p(vb), _p(pc), _p(ic), _p(vc))
    assert w >= 0
    return pc, ic[:w], vc[:w]


def daxpyi(a, x, indx, y):
    """y[indx] += a*x (level1/aoclsparse_axpyi.hpp:35-50).  Returns (status, y)."""
    x, indx, y = _f64(x), _i32(indx), _f64(y).copy()
    st = lib().orc_daxpyi(c_i32(len(indx)), c_dbl(a), _p(x), _p(indx), _p(y))
    return st, y


def ddoti(x, indx, y):
    x, indx, y = _f64(x), _i32(indx), _f64(y)
    lib().orc_ddoti.restype = c_dbl
    return lib().orc_ddoti(c_i32(len(indx)), _p(x), _p(indx), _p(y))


def droti(x, indx, y, c, s):
    x, indx, y = _f64(x).copy(), _i32(indx), _f64(y).copy()
    st = lib().orc_droti(c_i32(len(indx)), _p(x), _p(indx), _p(y), c_dbl(c), c_dbl(s))
    return st, x, y


def gthr(y, indx, zero=False):
    """x = y[indx]; gthrz also clears those entries (level1/aoclsparse_gthr.hpp:33-62).  Returns (x, y)."""
    y = np.array(y).copy()
    x = y[np.asarray(indx)].copy()
    if zero:
        y[np.asarray(indx)] = 0
    return x, y


def sctr(x, indx, y):
    """y[indx] = x (level1/aoclsparse_sctr.hpp:34-53)."""
    y = np.array(y).copy()
    y[np.asarray(indx)] = np.asarray(x)[: len(indx)]
    return y


def csr2dia(m, n, base, ptr, ind, val):
    """(ndiag, dia_offset, dia_val[m*ndiag]) -- convert.cpp:510-566, convert.hpp:291-387."""
    ptr, ind, val = _i32(ptr), _i32(ind), _f64(val)
    lib().orc_csr2dia_ndiag.restype = c_i32
    nd = lib().orc_csr2dia_ndiag(c_i32(m), c_i32(n), c_int(base), _p(ptr), _p(ind))
    off = np.zeros(max(nd, 1), dtype=np.int32)
    dv = np.zeros(max(nd * m, 1), dtype=np.float64)
    st = lib().orc_dcsr2dia(c_i32(m), c_i32(n), c_int(base), _p(ptr), _p(ind), _p(val), c_i32(nd), _p(off), _p(dv))
    assert st == 0
    return nd, off[:nd], dv[: nd * m]


def ddiamv(alpha, m, n, dia_val, dia_offset, x, beta, y):
    dia_val, dia_offset, x, y = _f64(dia_val), _i32(dia_offset), _f64(x), _f64(y).copy()
    lib().orc_ddiamv(c_dbl(alpha), c_i32(m), c_i32(n), _p(dia_val), _p(dia_offset), c_i32(len(dia_offset)), _p(x),
                     c_dbl(beta), _p(y))
    return y


def csr2bsr(m, n, base, ptr, ind, val, dim, rowmajor):
    """(bsr_row_ptr, bsr_col_ind, bsr_val) -- convert.cpp:596-729, convert.hpp:389-551."""
    ptr, ind, val = _i32(ptr), _i32(ind), _f64(val)
    mb = (m + dim - 1) // dim
    bp = np.zeros(mb + 1, dtype=np.int32)
    lib().orc_csr2bsr_nnz.restype = c_i32
    nb = lib().orc_csr2bsr_nnz(c_i32(m), c_i32(n), c_int(base), _p(ptr), _p(ind), c_i32(dim), _p(bp))
    assert nb >= 0
    bi = np.zeros(max(nb, 1), dtype=np.int32)
    bv = np.zeros(max(nb * dim * dim, 1), dtype=np.float64)
    st = lib().orc_dcsr2bsr(c_i32(m), c_i32(n), c_int(base), c_int(1 if rowmajor else 0), _p(val), _p(ptr), _p(ind),
                            c_i32(dim), _p(bv), _p(bp), _p(bi))
    assert st == 0
    return bp, bi[:nb], bv[: nb * dim * dim]


def dbsrmv(alpha, mb, dim, base, val, col, ptr, x, beta, y):
    val, col, ptr, x, y = _f64(val), _i32(col), _i32(ptr), _f64(x), _f64(y).copy()
    lib().orc_dbsrmv(c_dbl(alpha), c_i32(mb), c_i32(dim), c_int(base), _p(val), _p(col), _p(ptr), _p(x), c_dbl(beta), _p(y))
    return y


def dsorv(n, base, ptr, ind, val, omega, alpha, x, b):
    """One forward SOR sweep (solvers/aoclsparse_sorv.hpp:78-113); returns (status, x)."""
    ptr, ind, val, x, b = _i32(ptr), _i32(ind), _f64(val), _f64(x).copy(), _f64(b)
    st = lib().orc_dsorv(c_i32(n), c_int(base), _p(ptr), _p(ind), _p(val), c_dbl(omega), c_dbl(alpha), _p(x), _p(b))
    return st, x


# ---- complex CG / GMRES, numpy restatements of solvers/aoclsparse_itsol_functions.hpp:632-875 and :910-1367 for
# T = std::complex (dense operator A; no preconditioner).  No reference vectors exist for them: parity unpinned, the
# checks are exit status, iteration counts and the solver tolerances.
def zcg(A, b, x0, rtol, atol, maxit):
    """unconjugated products r.z and p.q, rz starts at (1, 1): the reference's complex CG.  -> (status, x, niter, rnorm)"""
    A, b, x = np.asarray(A, np.complex128), np.asarray(b, np.complex128), np.asarray(x0, np.complex128).copy()
    tiny = 0.02 * np.finfo(np.float64).eps
    brtol = rtol * np.linalg.norm(b)
    r = -b + A @ x
    p = np.zeros_like(x)
    rz, niter = 1 + 1j, 0
    rn = np.linalg.norm(r)
    while True:
        if (0 < atol and rn <= atol) or (0 < rtol and rn <= brtol):
            return 0, x, niter, rn
        if maxit > 0 and niter > maxit:
            return 7, x, niter, rn  # aoclsparse_status_maxit
        niter += 1
        z = r.copy()
        rz_new = np.sum(r * z)
        if abs(rz) <= tiny:
            return 8, x, niter, rn
        beta, rz = rz_new / rz, rz_new
        p = beta * p - z
        q = A @ p
        pq = np.sum(p * q)
        if abs(pq) <= tiny:
            return 8, x, niter, rn
        alpha = rz / pq
        x = x + alpha * p
        r = r + alpha * q
        rn = np.linalg.norm(r)


def zgmres(A, b, x0, m, rtol, atol, maxit):
    """-> (status, x, niter, rnorm).  The reference's control flow (restart cycles, convergence tested at the end of a cycle,
    early exit when the new direction vanishes) around the textbook complex Arnoldi / Givens steps: h(i,j) = v_i^H w,
    rotation [c s; -conj(s) c] from ?lartg on (h(j,j), |w|).  (The reference's own complex variant omits conjugations and
    diverges on general complex matrices; see itsol_api.cpp.)"""
    A, b, x = np.asarray(A, np.complex128), np.asarray(b, np.complex128), np.asarray(x0, np.complex128).copy()
    n, niter = len(b), 0
    brtol = rtol * np.linalg.norm(b)
    while True:
        V = np.zeros((m + 1, n), np.complex128)
        H = np.zeros((m, m), np.complex128)
        g, s, c = np.zeros(m + 1, np.complex128), np.zeros(m, np.complex128), np.zeros(m)
        V[0] = b - A @ x
        rn = np.linalg.norm(V[0])
        g[0] = rn
        if (0 < rn <= atol) or (0 < rn <= brtol) or rn == 0:
            return 0, x, niter, rn
        V[0] /= rn
        j = 0
        while j < m:
            w = A @ V[j]
            for i in range(j + 1):
                H[i, j] = np.sum(np.conj(V[i]) * w)
            for i in range(j + 1):
                w = w - H[i, j] * V[i]
            hh = np.linalg.norm(w)
            if hh < atol or hh < brtol:
                return 0, x, niter + j + 1, hh
            V[j + 1] = w / hh
            for i in range(j):
                r1, r2 = H[i, j], H[i + 1, j]
                H[i, j], H[i + 1, j] = c[i] * r1 + s[i] * r2, -np.conj(s[i]) * r1 + c[i] * r2
            f, gg = H[j, j], hh + 0j
            if f == 0:
                c[j], s[j], H[j, j] = 0.0, np.conj(gg) / abs(gg), abs(gg)
            else:
                f2, g2 = abs(f) ** 2, abs(gg) ** 2
                c[j] = np.sqrt(f2 / (f2 + g2))
                H[j, j] = f / c[j]
                s[j] = np.conj(gg) * (f / np.sqrt(f2 * (f2 + g2)))
            g0 = g[j]
            g[j], g[j + 1] = c[j] * g0, -np.conj(s[j]) * g0
            j += 1
        y = np.zeros(m, np.complex128)
        for jj in range(m - 1, -1, -1):
            y[jj] = (g[jj] - np.sum(H[jj, jj + 1:m] * y[jj + 1:m])) / H[jj, jj]
        x = x + y @ V[:m]
        rn = abs(g[m])
        niter += m
        below = (0 < atol and rn <= atol) or (0 < rn <= brtol)
        if below:
            return 0, x, niter, rn
        if maxit > 0 and niter >= maxit:
            return 7, x, niter, rn
