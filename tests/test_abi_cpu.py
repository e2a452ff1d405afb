"""CPU tier: the C-ABI library loads, exports every symbol include/*.h declares, and its host-side
logic (descriptor, create/validate, hints, optimize -> clean CSR, status codes) matches the reference's
behaviour and the oracle.  No compute entry point is exercised here (needs a GPU)."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

import oracle
from util import ROOT, pkg, random_csr

P = pkg()
L = P.lib()


def _declared_symbols():
    names = []
    for h in ("aoclsparse.h", "aoclsparse_mi355.h"):
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        src = re.sub(r"#define DLL_PUBLIC.*", "", src)
        for m in re.finditer(r"DLL_PUBLIC\s+[^;(]*?\b(\w+)\s*\(", src):
            names.append(m.group(1))
    return names


def test_every_declared_symbol_is_exported_and_bound():
    names = _declared_symbols()
    assert len(names) >= 60
    out = subprocess.check_output(["nm", "-D", "--defined-only", P.LIB_PATH], text=True)
    exported = {ln.split()[-1] for ln in out.splitlines() if " T " in ln}
    for n in names:
        assert n in exported, "declared in include/ but not exported: " + n
        assert n in P.SIGNATURES, "no ctypes signature for " + n
    # nothing but the C ABI and the reference's sixteen C++ template instantiations leaks out (library built with
    # -fvisibility=hidden); template instantiations are weak definitions ("W")
    cxx = [ln.strip() for ln in open(os.path.join(ROOT, "tests", "golden", "cxx_symbols.txt")) if ln.strip()]
    assert len(cxx) == 16
    weak = {ln.split()[-1] for ln in out.splitlines() if " W " in ln}
    for n in cxx:  # mv / trsv / sp2m / create_csr x s / d / c / z (aoclsparse_mv.cpp:351-360, trsv.cpp:419-431, csr2m.cpp:863-873,
        assert n in weak or n in exported, "C++ entry point of the reference not exported: " + n  # create.cpp:99-110)
    leaked = [s for s in exported | weak if not (s.startswith("aoclsparse_") or s.startswith("mi355_") or s in cxx)]
    assert not leaked, leaked


def test_soname_and_versioned_names():
    """library/CMakeLists.txt:144-145: SOVERSION = VERSION = 5.3.2, so a program linked against the reference asks the loader for
    libaoclsparse.so.5.3.2; this library carries that SONAME and the directory holds the name chain .so -> .so.5 -> .so.5.3.2."""
    dyn = subprocess.check_output(["readelf", "-d", P.LIB_PATH], text=True)
    assert "Library soname: [libaoclsparse.so.5.3.2]" in dyn, dyn
    d = os.path.dirname(P.LIB_PATH)
    for name in ("libaoclsparse.so", "libaoclsparse.so.5", "libaoclsparse.so.5.3.2"):
        assert os.path.samefile(os.path.join(d, name), P.LIB_PATH), name


def test_library_loads_on_the_older_hip_runtime_a_framework_bundles():
    """A process that imports torch first runs on torch's bundled libamdhip64 (ROCm 7.0), not /opt/rocm's 7.2: the library may
    only bind HIP symbols that runtime has (symbol versions up to hip_6.x; newer entry points such as hipStreamGetId are looked up
    with dlsym at run time).  Round 6 found out on the GPU box."""
    syms = subprocess.check_output(["nm", "-D", "--with-symbol-versions", P.LIB_PATH], text=True)
    vers = sorted({ln.rsplit("@", 1)[1] for ln in syms.splitlines() if " U " in ln and "@hip_" in ln})
    assert vers and all(tuple(int(t) for t in v[4:].split(".")) < (7, 0) for v in vers), vers


def test_declaration_only_cxx_program_links(tmp_path):
    """A C++ program that only DECLARES aoclsparse::mv / create_csr (what the reference's header gives it) links against this
    library through the versioned name and gets the library's status codes back without a device call (null-pointer checks)."""
    src = tmp_path / "decl_only.cpp"
    src.write_text("""
#include "aoclsparse.h"
#include <complex>
namespace aoclsparse {
template <typename T> aoclsparse_status mv(aoclsparse_operation, const T *, aoclsparse_matrix, const aoclsparse_mat_descr,
                                           const T *, const T *, T *);
template <typename T> aoclsparse_status create_csr(aoclsparse_matrix *, aoclsparse_index_base, aoclsparse_int, aoclsparse_int,
                                                   aoclsparse_int, aoclsparse_int *, aoclsparse_int *, T *, bool = false);
template <typename T> aoclsparse_status trsv(const aoclsparse_operation, const T, aoclsparse_matrix, const aoclsparse_mat_descr,
                                             const T *, const aoclsparse_int, T *, const aoclsparse_int, aoclsparse_int = -1);
}
int main() {
    double a = 1.0, x[1] = {1.0}, y[1] = {0.0};
    if(aoclsparse::mv<double>(aoclsparse_operation_none, &a, nullptr, nullptr, x, &a, y) != aoclsparse_status_invalid_pointer) return 1;
    aoclsparse_int rp[2] = {0, 1}, ci[1] = {0};
    std::complex<float> v[1] = {{1.f, 2.f}};
    aoclsparse_matrix A = nullptr;
    if(aoclsparse::create_csr<std::complex<float>>(&A, aoclsparse_index_base_zero, 1, 1, 1, rp, ci, v) != aoclsparse_status_success) return 2;
    if(aoclsparse::trsv<std::complex<float>>(aoclsparse_operation_none, v[0], A, nullptr, v, 1, v, 1) != aoclsparse_status_invalid_pointer) return 3;
    return aoclsparse_destroy(&A) == aoclsparse_status_success ? 0 : 4;
}
""")
    exe = tmp_path / "decl_only"
    d = os.path.dirname(P.LIB_PATH)
    # (a sanitizer build of the library -- tests/run_san.sh -- leaves its __asan_* / __ubsan_* references to the preloaded runtime)
    san = ["-Wl,--allow-shlib-undefined"] if os.path.basename(d) == "lib_san" else []
    subprocess.check_call(["g++", "-std=c++17", str(src), "-I" + os.path.join(ROOT, "include"), "-L" + d, "-laoclsparse",
                           "-Wl,-rpath," + d, "-o", str(exe)] + san)
    needed = subprocess.check_output(["readelf", "-d", str(exe)], text=True)
    assert "libaoclsparse.so.5.3.2" in needed
    assert subprocess.call([str(exe)]) == 0


def test_product_does_not_reference_the_oracle():
    out = subprocess.check_output(["nm", "-D", P.LIB_PATH], text=True)
    assert "orc_" not in out
    ldd = subprocess.check_output(["ldd", P.LIB_PATH], text=True)
    assert "liboracle" not in ldd and "amdhip64" in ldd
    for root, _, files in os.walk(os.path.join(ROOT, "aocl-sparse_amd")):
        for f in files:
            if f.endswith((".cpp", ".hip", ".hpp", ".py")):
                txt = open(os.path.join(root, f)).read()
                assert "import oracle" not in txt and "liboracle" not in txt, f


def test_version_and_context_shims():
    assert b"5.3.2" in L.aoclsparse_get_version()
    assert L.aoclsparse_is_avx512_build() == 0
    assert L.aoclsparse_enable_instructions(b"AVX2") == 0
    assert L.aoclsparse_enable_instructions(b"bogus") == 5
    assert L.aoclsparse_enable_instructions(None) == 2


def test_hip_runtime_path_names_the_runtime_the_library_is_bound_to():
    """a process can hold two HIP runtimes (a framework's bundled copy next to /opt/rocm's); streams given to
    aoclsparse_mi355_set_stream must come from the one this names.  No device is touched."""
    import ctypes
    st, path = P.hip_runtime_path()
    assert st == 0 and "libamdhip64" in os.path.basename(path) and os.path.exists(path)
    with open("/proc/self/maps") as f:
        assert any(os.path.realpath(path) == os.path.realpath(ln.split()[-1]) for ln in f if "libamdhip64" in ln)
    assert L.aoclsparse_mi355_hip_runtime_path(None, 10) == 2  # invalid_pointer
    buf = ctypes.create_string_buffer(8)
    assert L.aoclsparse_mi355_hip_runtime_path(buf, 0) == 3  # invalid_size
    assert L.aoclsparse_mi355_hip_runtime_path(buf, 8) == 0 and len(buf.value) == 7 and path.startswith(buf.value.decode())


def test_descriptor_api():
    # library/src/extra/aoclsparse_auxiliary.cpp:191-360
    d = ctypes.c_void_p()
    assert L.aoclsparse_create_mat_descr(None) == 2
    assert L.aoclsparse_create_mat_descr(ctypes.byref(d)) == 0
    assert L.aoclsparse_get_mat_index_base(d) == 0 and L.aoclsparse_get_mat_type(d) == 0
    assert L.aoclsparse_get_mat_fill_mode(d) == 0 and L.aoclsparse_get_mat_diag_type(d) == 0
    assert L.aoclsparse_set_mat_index_base(d, 1) == 0 and L.aoclsparse_get_mat_index_base(d) == 1
    assert L.aoclsparse_set_mat_index_base(d, 2) == 5
    assert L.aoclsparse_set_mat_type(d, 3) == 0 and L.aoclsparse_get_mat_type(d) == 3
    assert L.aoclsparse_set_mat_type(d, 4) == 5
    assert L.aoclsparse_set_mat_fill_mode(d, 1) == 0 and L.aoclsparse_set_mat_fill_mode(d, 2) == 5
    assert L.aoclsparse_set_mat_diag_type(d, 2) == 0 and L.aoclsparse_set_mat_diag_type(d, 3) == 5
    assert L.aoclsparse_set_mat_type(None, 0) == 2
    assert L.aoclsparse_get_mat_type(None) == 0 and L.aoclsparse_get_mat_diag_type(None) == 0
    d2 = ctypes.c_void_p()
    assert L.aoclsparse_create_mat_descr(ctypes.byref(d2)) == 0
    assert L.aoclsparse_copy_mat_descr(d2, d) == 0 and L.aoclsparse_get_mat_type(d2) == 3
    assert L.aoclsparse_copy_mat_descr(None, d) == 2 and L.aoclsparse_copy_mat_descr(d2, None) == 2
    assert L.aoclsparse_destroy_mat_descr(d) == 0 and L.aoclsparse_destroy_mat_descr(d2) == 0
    assert L.aoclsparse_destroy_mat_descr(None) == 0


def test_create_validation_matches_oracle():
    # create/aoclsparse_create.cpp:34-97 -> aoclsparse_mat_check_internal
    cases = [
        (5, 5, [0, 2, 3, 4, 7, 8], [0, 3, 1, 2, 1, 3, 4, 4], 0),
        (2, 2, [0, 1, 2], [0, 2], 0),          # column out of range
        (2, 2, [0, 2, 3], [0, 0, 1], 0),       # duplicate diagonal
        (2, 2, [1, 1, 2], [0, 1], 0),          # ptr[0] != base
        (2, 2, [0, 2, 1], [0, 1], 0),          # decreasing ptr / ptr[m] != nnz
        (3, 3, [1, 2, 3, 4], [1, 2, 3], 1),    # base one, fine
    ]
    for m, n, rp, ci, base in cases:
        nnz = len(ci)
        v = np.ones(max(nnz, 1))
        A = P.Matrix(base, m, n, rp, ci, v[:nnz] if nnz else v)
        A.nnz = nnz
        h = ctypes.c_void_p()
        st = L.aoclsparse_create_dcsr(ctypes.byref(h), base, m, n, nnz, P._ptr(A.row_ptr), P._ptr(A.col_ind),
                                      P._ptr(A.val))
        so, _, _ = oracle.mat_check(m, n, nnz, rp, ci, v, 0, base)
        assert st == so, (rp, ci, st, so)
        if st == 0:
            assert L.aoclsparse_destroy(ctypes.byref(h)) == 0 and not h.value
    h = ctypes.c_void_p()
    assert L.aoclsparse_create_dcsr(None, 0, 1, 1, 1, None, None, None) == 2
    assert L.aoclsparse_create_dcsr(ctypes.byref(h), 0, 1, 1, 1, None, None, None) == 2
    a = np.zeros(2, np.int32)
    assert L.aoclsparse_create_dcsr(ctypes.byref(h), 0, -1, 1, 0, P._ptr(a), P._ptr(a), P._ptr(a)) == 3
    assert L.aoclsparse_destroy(None) == 0


def test_optimize_produces_reference_clean_csr(kats):
    # tests/unit_tests/hint_tests.cpp:75-170 via aoclsparse_set_sv_hint + aoclsparse_optimize
    for c in kats["clean_csr"]:
        A = P.Matrix(0, c["m"], c["n"], c["row_ptr"], c["col_ind"], np.array(c["val"], np.float64))
        assert A.status == 0, c["name"]
        d = P.Descr(mtype=P.TYPE_TRIANGULAR)
        assert L.aoclsparse_set_sv_hint(A.h, P.OP_NONE, d.h, 1) == 0
        assert L.aoclsparse_optimize(A.h) == 0, c["name"]
        e, g = A.export(), A.export_diag()
        x = c["exp"]
        assert e["status"] == 0 and g["status"] == 0
        assert np.array_equal(e["row_ptr"], x["icrow"]), c["name"]
        assert np.array_equal(e["col_ind"], x["icol"]), c["name"]
        assert np.array_equal(e["val"], np.array(x["aval"], np.float64)), c["name"]
        dim = min(c["m"], c["n"])
        assert np.array_equal(g["idiag"][:dim], x["idiag"]), c["name"]
        assert np.array_equal(g["iurow"][:dim], x["iurow"]), c["name"]
        assert g["is_internal"] == x["is_internal"] and e["aliased"] == (not x["is_internal"])


@pytest.mark.parametrize("base", [0, 1])
def test_optimize_random_matches_oracle_bit_exact(base):
    rp, ci, v = random_csr(7 + base, 300, 280, lambda r, i: r.integers(0, 12), base=base, sort=False)
    # knock out some diagonals / keep others so fill-in paths are exercised
    A = P.Matrix(base, 300, 280, rp, ci, v)
    assert A.status == 0
    d = P.Descr(base=base, mtype=P.TYPE_TRIANGULAR)
    assert L.aoclsparse_set_sv_hint(A.h, P.OP_NONE, d.h, 3) == 0
    assert L.aoclsparse_optimize(A.h) == 0
    e, g = A.export(), A.export_diag()
    o = oracle.dcsr_optimize(300, 280, len(v), base, rp, ci, v)
    assert o["status"] == 0 and e["base"] == o["base"]
    assert np.array_equal(e["row_ptr"], o["ptr"]) and np.array_equal(e["col_ind"], o["ind"])
    assert np.array_equal(e["val"], o["val"])
    assert np.array_equal(g["idiag"], o["idiag"]) and np.array_equal(g["iurow"], o["iurow"])
    assert g["is_internal"] == o["is_internal"]


def test_export_before_optimize_returns_user_arrays():
    rp, ci, v = random_csr(3, 20, 20, lambda r, i: 3)
    A = P.Matrix(0, 20, 20, rp, ci, v)
    e = A.export()
    assert e["status"] == 0 and e["aliased"] and e["nnz"] == len(v)
    assert A.export_diag()["status"] == 12  # invalid_operation: no clean CSR yet
    As = P.Matrix(0, 20, 20, rp, ci, v.astype(np.float32))
    base, m, n, nnz = ctypes.c_int(), ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
    a, b, c = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
    st = L.aoclsparse_export_dcsr(As.h, ctypes.byref(base), ctypes.byref(m), ctypes.byref(n),
                                  ctypes.byref(nnz), ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
    assert st == 9  # wrong_type
    assert L.aoclsparse_export_dcsr(None, None, None, None, None, None, None, None) == 2


def test_hint_validation():
    # analysis/aoclsparse_analysis.cpp:568-625, tests/unit_tests/hint_tests.cpp:253-357
    rp, ci, v = random_csr(1, 10, 10, lambda r, i: 2)
    A = P.Matrix(0, 10, 10, rp, ci, v)
    d = P.Descr()
    assert L.aoclsparse_set_mv_hint(None, P.OP_NONE, d.h, 1) == 2
    assert L.aoclsparse_set_mv_hint(A.h, P.OP_NONE, None, 1) == 2
    assert L.aoclsparse_set_mv_hint(A.h, 114, d.h, 1) == 5
    assert L.aoclsparse_set_mv_hint(A.h, P.OP_NONE, d.h, -1) == 5
    assert L.aoclsparse_set_mv_hint(A.h, P.OP_NONE, d.h, 0) == 5          # nop == 0 needs a kid
    assert L.aoclsparse_set_mv_hint_kid(A.h, P.OP_NONE, d.h, 0, 2) == 0
    d1 = P.Descr(base=1)
    assert L.aoclsparse_set_mv_hint(A.h, P.OP_NONE, d1.h, 1) == 5         # base mismatch
    for fn in (L.aoclsparse_set_sv_hint, L.aoclsparse_set_mm_hint, L.aoclsparse_set_2m_hint):
        assert fn(A.h, P.OP_TRANSPOSE, d.h, 7) == 0
    assert L.aoclsparse_set_memory_hint(A.h, P.MEM_MINIMAL) == 0
    assert L.aoclsparse_set_memory_hint(A.h, 7) == 5
    assert L.aoclsparse_set_memory_hint(None, 0) == 2
    assert L.aoclsparse_optimize(None) == 2
    assert L.aoclsparse_optimize(A.h) == 0  # minimal memory: host analysis only, no device copies


def test_argument_checks_that_precede_any_device_work():
    """Status codes pinned by csrmv_tests.cpp:33-183, mv_tests.cpp:56-341, trsv_tests.cpp negative
    cases and csrmm_tests.cpp:1833-1995; all of them return before the GPU is touched."""
    rp, ci, v = random_csr(2, 6, 6, lambda r, i: 2)
    nnz = len(v)
    x, y = np.ones(6), np.zeros(6)
    d = P.Descr()
    one, zero = ctypes.c_double(1.0), ctypes.c_double(0.0)
    f = L.aoclsparse_dcsrmv
    args = lambda **k: [k.get("op", 111), k.get("alpha", ctypes.byref(one)), k.get("m", 6), k.get("n", 6),
                        k.get("nnz", nnz), k.get("val", P._ptr(v)), k.get("col", P._ptr(ci)),
                        k.get("row", P._ptr(rp)), k.get("descr", d.h), k.get("x", P._ptr(x)),
                        k.get("beta", ctypes.byref(zero)), k.get("y", P._ptr(y))]
    assert f(*args(alpha=None)) == 2 and f(*args(beta=None)) == 2 and f(*args(descr=None)) == 2
    assert f(*args(op=114)) == 5
    assert f(*args(m=-1)) == 3 and f(*args(n=-1)) == 3 and f(*args(nnz=-1)) == 3
    for k in ("val", "col", "row", "x", "y"):
        assert f(*args(**{k: None})) == 2
    dt = P.Descr(mtype=P.TYPE_TRIANGULAR)
    assert f(*args(descr=dt.h)) == 1           # raw csrmv: only general + symmetric (csrmv.hpp:83-88)
    ds = P.Descr(mtype=P.TYPE_SYMMETRIC)
    assert f(*args(descr=ds.h, m=5)) == 3      # symmetric must be square
    A = P.Matrix(0, 6, 6, rp, ci, v)
    g = L.aoclsparse_dmv
    assert g(111, None, A.h, d.h, P._ptr(x), ctypes.byref(zero), P._ptr(y)) == 2
    assert g(111, ctypes.byref(one), None, d.h, P._ptr(x), ctypes.byref(zero), P._ptr(y)) == 2
    assert g(111, ctypes.byref(one), A.h, None, P._ptr(x), ctypes.byref(zero), P._ptr(y)) == 2
    assert g(111, ctypes.byref(one), A.h, d.h, None, ctypes.byref(zero), P._ptr(y)) == 2
    assert g(114, ctypes.byref(one), A.h, d.h, P._ptr(x), ctypes.byref(zero), P._ptr(y)) == 5
    d1 = P.Descr(base=1)
    assert g(111, ctypes.byref(one), A.h, d1.h, P._ptr(x), ctypes.byref(zero), P._ptr(y)) == 5
    assert L.aoclsparse_smv(111, ctypes.byref(one), A.h, d.h, P._ptr(x), ctypes.byref(zero), P._ptr(y)) == 9
    # trsv (trsv.cpp:59-113)
    t = L.aoclsparse_dtrsv
    assert t(111, 1.0, None, dt.h, P._ptr(x), P._ptr(y)) == 2
    assert t(111, 1.0, A.h, dt.h, None, P._ptr(y)) == 2
    assert t(111, 1.0, A.h, d.h, P._ptr(x), P._ptr(y)) == 5        # general type is invalid for trsv
    dz = P.Descr(mtype=P.TYPE_TRIANGULAR, diag=P.DIAG_ZERO)
    assert t(111, 1.0, A.h, dz.h, P._ptr(x), P._ptr(y)) == 5
    assert L.aoclsparse_strsv(111, 1.0, A.h, dt.h, P._ptr(x), P._ptr(y)) == 9
    assert L.aoclsparse_dtrsv_strided(111, 1.0, A.h, dt.h, P._ptr(x), 0, P._ptr(y), 1) == 5
    R = P.Matrix(0, 6, 7, rp, ci, v)
    assert t(111, 1.0, R.h, dt.h, P._ptr(x), P._ptr(y)) == 5       # not square
    # a matrix without a full diagonal cannot be solved with a non-unit diagonal (trsv.cpp:133-137)
    H = P.Matrix(0, 3, 3, [0, 1, 1, 2], [0, 2], np.ones(2))
    assert t(111, 1.0, H.h, dt.h, P._ptr(x), P._ptr(y)) == 5
    assert L.aoclsparse_dtrsv_kid(111, 1.0, A.h, P.Descr(mtype=3, diag=P.DIAG_UNIT).h, P._ptr(x), P._ptr(y), 4) == 14
    # csrmm (csrmm.hpp:448-611)
    B, C = np.ones(36), np.zeros(36)
    mm = L.aoclsparse_dcsrmm
    assert mm(111, 1.0, None, d.h, 0, P._ptr(B), 6, 6, 0.0, P._ptr(C), 6) == 2
    assert mm(111, 1.0, A.h, d.h, 0, None, 6, 6, 0.0, P._ptr(C), 6) == 2
    assert mm(114, 1.0, A.h, d.h, 0, P._ptr(B), 6, 6, 0.0, P._ptr(C), 6) == 5
    assert mm(111, 1.0, A.h, d.h, 2, P._ptr(B), 6, 6, 0.0, P._ptr(C), 6) == 5
    assert mm(111, 1.0, A.h, dt.h, 0, P._ptr(B), 6, 6, 0.0, P._ptr(C), 6) == 1
    assert mm(111, 1.0, A.h, d.h, 0, P._ptr(B), 6, 5, 0.0, P._ptr(C), 6) == 3   # ldb too small
    assert mm(111, 1.0, A.h, d.h, 1, P._ptr(B), 6, 6, 0.0, P._ptr(C), 5) == 3   # ldc too small
    assert mm(111, 1.0, A.h, d.h, 0, P._ptr(B), -1, 6, 0.0, P._ptr(C), 6) == 3
    assert mm(111, 0.0, A.h, d.h, 0, P._ptr(B), 6, 6, 1.0, P._ptr(C), 6) == 0   # alpha=0,beta=1: no-op
    assert mm(111, 1.0, A.h, d.h, 0, P._ptr(B), 0, 6, 0.0, P._ptr(C), 6) == 0   # n == 0 quick return
    assert L.aoclsparse_scsrmm(111, 1.0, A.h, d.h, 0, P._ptr(B), 6, 6, 0.0, P._ptr(C), 6) == 9
    assert mm(111, 1.0, A.h, d.h, 0, P._ptr(B), 6, 2 ** 30, 0.0, P._ptr(C), 6) == 3  # dim*ld overflows int32
    # sp2m family (csr2m.cpp:592-740): checks first; a valid product needs the GPU
    Cc = ctypes.c_void_p()
    assert L.aoclsparse_sp2m(111, d.h, A.h, 111, d.h, None, 2, ctypes.byref(Cc)) == 2
    assert L.aoclsparse_sp2m(111, None, A.h, 111, d.h, A.h, 2, ctypes.byref(Cc)) == 2
    assert L.aoclsparse_sp2m(111, d.h, R.h, 111, d.h, A.h, 2, ctypes.byref(Cc)) == 3  # 6x7 times 6x6
    assert L.aoclsparse_sp2m(114, d.h, A.h, 111, d.h, A.h, 2, ctypes.byref(Cc)) == 5
    assert L.aoclsparse_sp2m(111, d1.h, A.h, 111, d.h, A.h, 2, ctypes.byref(Cc)) == 5  # base mismatch
    assert L.aoclsparse_sp2m(111, dt.h, A.h, 111, d.h, A.h, 2, ctypes.byref(Cc)) == 1  # general only
    assert L.aoclsparse_scsr2m(111, d.h, A.h, 111, d.h, A.h, 2, ctypes.byref(Cc)) == 9
    assert L.aoclsparse_sp2m(111, d.h, A.h, 111, d.h, R.h, 2, ctypes.byref(Cc)) in (0, 4)  # 4: no HIP device here
    # empty product: an empty C is still allocated (csr2m.cpp:705-735), no device needed
    E = P.Matrix(0, 6, 6, np.zeros(7, np.int32), np.zeros(1, np.int32), np.zeros(1))
    E.nnz = 0
    Eh = ctypes.c_void_p()
    assert L.aoclsparse_create_dcsr(ctypes.byref(Eh), 0, 6, 6, 0, P._ptr(E.row_ptr), P._ptr(E.col_ind), P._ptr(E.val)) == 0
    assert L.aoclsparse_sp2m(111, d.h, Eh, 111, d.h, A.h, 2, ctypes.byref(Cc)) == 0 and Cc.value
    base, mm_, nn_, nz = ctypes.c_int(), ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
    a, b, c = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
    assert L.aoclsparse_export_dcsr(Cc, ctypes.byref(base), ctypes.byref(mm_), ctypes.byref(nn_), ctypes.byref(nz),
                                    ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)) == 0
    assert (mm_.value, nn_.value, nz.value, base.value) == (6, 6, 0, 0)
    assert L.aoclsparse_destroy(ctypes.byref(Cc)) == 0 and L.aoclsparse_destroy(ctypes.byref(Eh)) == 0


def test_row_block_planner_properties():
    """mi355_csrmv_plan_host: blocks tile [0,m) with whole rows, <= tile nnz unless a single long row."""
    rp, ci, v = random_csr(5, 3000, 3000, lambda r, i: 5000 if i % 701 == 3 else r.integers(0, 30))
    m = 3000
    for tile in (512, 1024, 2048):
        blk = np.zeros(L.mi355_csrmv_plan_bound(m, len(v)), dtype=np.int32)
        nb = L.mi355_csrmv_plan_host(m, 0, tile, P._ptr(rp), P._ptr(blk))
        assert nb > 0
        blk = blk[: 2 * (nb + 1)].reshape(nb + 1, 2)
        rows, pos = blk[:, 0], blk[:, 1]
        assert rows[0] == 0 and rows[nb] == m and pos[0] == 0 and pos[nb] == len(v)
        assert np.all(np.diff(rows) > 0) and np.array_equal(pos, rp[rows])
        cnt, nr = np.diff(pos), np.diff(rows)
        assert np.all((cnt <= tile) | (nr == 1)) and np.all(nr <= min(512, tile // 2))
        assert np.sum(cnt > tile) == 5
    assert L.mi355_csrmv_plan_host(-1, 0, 1024, P._ptr(rp), P._ptr(blk)) < 0
    assert L.mi355_csrmv_plan_host(m, 0, 256, P._ptr(rp), P._ptr(blk)) < 0
    # base-1 row_ptr gives the same 0-based plan
    b1 = np.zeros(L.mi355_csrmv_plan_bound(m, len(v)), dtype=np.int32)
    rp1 = (rp + 1).astype(np.int32)
    nb1 = L.mi355_csrmv_plan_host(m, 1, 2048, P._ptr(rp1), P._ptr(b1))
    assert nb1 == nb and np.array_equal(b1[: 2 * (nb + 1)].reshape(nb + 1, 2), blk)


def test_value_mutation_and_copy_host_semantics():
    """aoclsparse_?set_value / ?update_values / copy (auxiliary.hpp:216-270, 388-470; auxiliary.cpp:775-835):
    writes go through to the aliased user arrays, the clean copy is dropped, copy is deep."""
    rp = np.array([0, 2, 3, 4, 6, 7], np.int32)
    ci = np.array([3, 0, 1, 2, 1, 4, 4], np.int32)  # N5_1_hole: unsorted + a missing diagonal
    v = np.array([2, 1, 3, 4, 5, 7, 8], np.float64)
    A = P.Matrix(0, 5, 5, rp, ci, v)
    d = P.Descr(mtype=P.TYPE_TRIANGULAR)
    assert L.aoclsparse_set_sv_hint(A.h, P.OP_NONE, d.h, 1) == 0 and L.aoclsparse_optimize(A.h) == 0
    assert A.export()["nnz"] == 8  # clean copy with the inserted zero diagonal
    assert L.aoclsparse_dset_value(A.h, 3, 1, -5.5) == 0
    assert A.val[4] == -5.5  # user's array was written
    e = A.export()
    assert e["aliased"] and e["nnz"] == 7  # clean copy dropped, user arrays exported again
    assert L.aoclsparse_dset_value(A.h, 3, 3, 1.0) == 6   # (3,3) is not stored: invalid_index_value
    assert L.aoclsparse_dset_value(A.h, 5, 0, 1.0) == 5 and L.aoclsparse_dset_value(A.h, 0, -1, 1.0) == 5
    assert L.aoclsparse_sset_value(A.h, 0, 0, 1.0) == 9 and L.aoclsparse_dset_value(None, 0, 0, 1.0) == 2
    nv = np.arange(1.0, 8.0)
    assert L.aoclsparse_dupdate_values(A.h, 7, P._ptr(nv)) == 0 and np.array_equal(A.val, nv)
    assert L.aoclsparse_dupdate_values(A.h, 6, P._ptr(nv)) == 3 and L.aoclsparse_dupdate_values(A.h, 7, None) == 2
    assert L.aoclsparse_optimize(A.h) == 0  # hints were already consumed: no-op, like the reference
    C = ctypes.c_void_p()
    assert L.aoclsparse_copy(A.h, d.h, ctypes.byref(C)) == 0 and C.value and C.value != A.h.value
    base, m, n, nnz = ctypes.c_int(), ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
    a, b, c = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
    assert L.aoclsparse_export_dcsr(C, ctypes.byref(base), ctypes.byref(m), ctypes.byref(n), ctypes.byref(nnz),
                                    ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)) == 0
    assert a.value != A.row_ptr.ctypes.data and nnz.value == 7
    cv = np.ctypeslib.as_array(ctypes.cast(c, ctypes.POINTER(ctypes.c_double)), (7,)).copy()
    assert np.array_equal(cv, nv)
    assert L.aoclsparse_copy(None, d.h, ctypes.byref(C)) == 2
    assert L.aoclsparse_destroy(ctypes.byref(C)) == 0
    # trsm / dotmv argument checks (trsm.hpp:54-140, dotmv.hpp:42-46)
    x, y = np.ones(5), np.zeros(5)
    assert L.aoclsparse_ddotmv(111, 1.0, A.h, d.h, P._ptr(x), 0.0, P._ptr(y), None) == 2
    assert L.aoclsparse_ddotmv(111, 1.0, None, d.h, P._ptr(x), 0.0, P._ptr(y), P._ptr(y)) == 2
    Bm, Xm = np.ones(10), np.zeros(10)
    t = L.aoclsparse_dtrsm
    assert t(111, 1.0, None, d.h, 0, P._ptr(Bm), 2, 2, P._ptr(Xm), 2) == 2
    assert t(111, 1.0, A.h, P.Descr().h, 0, P._ptr(Bm), 2, 2, P._ptr(Xm), 2) == 5   # general type
    assert t(111, 1.0, A.h, d.h, 0, P._ptr(Bm), -1, 2, P._ptr(Xm), 2) == 3
    assert t(111, 1.0, A.h, d.h, 0, P._ptr(Bm), 0, 2, P._ptr(Xm), 2) == 0           # n == 0 quick return
    assert t(111, 1.0, A.h, d.h, 2, P._ptr(Bm), 2, 2, P._ptr(Xm), 2) == 5           # bad order
    assert t(114, 1.0, A.h, d.h, 0, P._ptr(Bm), 2, 2, P._ptr(Xm), 2) == 5
    assert t(111, 1.0, A.h, d.h, 0, P._ptr(Bm), 2, 2, P._ptr(Xm), 2) == 5           # missing diagonal, non-unit
    assert L.aoclsparse_strsm(111, 1.0, A.h, d.h, 0, P._ptr(Bm), 2, 2, P._ptr(Xm), 2) == 9


def test_remaining_hint_setters_and_ilu_hint():
    """analysis.cpp:644-731: dotmv / lu_smoother / sm / symgs / sorv hints share set_hint's checks; sm and sorv
    validate their extra enum first."""
    rp = np.array([0, 2, 3, 4, 7, 8], np.int32)
    ci = np.array([0, 3, 1, 2, 1, 3, 4, 4], np.int32)
    v = np.arange(1.0, 9.0)
    A, d = P.Matrix(0, 5, 5, rp, ci, v), P.Descr()
    assert L.aoclsparse_set_dotmv_hint(A.h, P.OP_NONE, d.h, 3) == 0
    assert L.aoclsparse_set_symgs_hint(A.h, P.OP_NONE, d.h, 3) == 0
    assert L.aoclsparse_set_lu_smoother_hint(A.h, P.OP_NONE, d.h, 3) == 0
    assert L.aoclsparse_set_sm_hint(A.h, P.OP_NONE, d.h, P.ORDER_ROW, 3) == 0
    assert L.aoclsparse_set_sm_hint(A.h, P.OP_NONE, d.h, 7, 3) == 5
    assert L.aoclsparse_set_sorv_hint(A.h, d.h, 2, 3) == 0 and L.aoclsparse_set_sorv_hint(A.h, d.h, 3, 3) == 5
    assert L.aoclsparse_set_symgs_hint(A.h, P.OP_NONE, d.h, -1) == 5
    assert L.aoclsparse_set_dotmv_hint(None, P.OP_NONE, d.h, 1) == 2
    assert L.aoclsparse_optimize(A.h) == 0


def _rand_rows(seed, m, n, maxlen):
    rng = np.random.default_rng(seed)
    lens = rng.integers(0, maxlen, m)
    lens[min(5, m - 1)] = 0
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    ci = np.concatenate([np.sort(rng.choice(n, k, replace=False)) for k in lens] + [np.zeros(0, np.int64)]).astype(np.int32)
    return rp, ci, rng.uniform(-1, 1, len(ci))


@pytest.mark.parametrize("base", [0, 1])
def test_ell_conversions_match_the_oracle(base, kats):
    """conversion/aoclsparse_convert.{cpp,hpp}: widths, ELL (-1 padding), ELLT (last-column padding), ELLT-HYB
    (row map, ell_m) -- int-exact against the restatement, which is pinned on ellmv_tests.cpp's 3x3."""
    import oracle
    m, n = 200, 180
    rp, ci, v = _rand_rows(41, m, n, 19)
    rp, ci = rp + base, ci + base
    d = P.Descr(base=base)
    w = ctypes.c_int32(-1)
    assert L.aoclsparse_csr2ell_width(m, len(v), P._ptr(rp), ctypes.byref(w)) == 0
    for layout, fn in (("ell", L.aoclsparse_dcsr2ell), ("ellt", L.aoclsparse_dcsr2ellt)):
        wo, ec, ev = oracle.csr2ell(layout, m, base, rp, ci, v)
        assert w.value == wo
        gc, gv = np.full(m * wo, 77, np.int32), np.full(m * wo, 77.0)
        assert fn(m, d.h, P._ptr(rp), P._ptr(ci), P._ptr(v), P._ptr(gc), P._ptr(gv), wo) == 0
        assert np.array_equal(gc, ec) and np.array_equal(gv, ev)
    wh, em = ctypes.c_int32(-1), ctypes.c_int32(-1)
    assert L.aoclsparse_csr2ellthyb_width(m, len(v), P._ptr(rp), ctypes.byref(em), ctypes.byref(wh)) == 0
    wo, emo, mp, hc, hv = oracle.csr2ell("hyb", m, base, rp, ci, v)
    assert (wh.value, em.value) == (wo, emo)
    gc, gv, gm, em2 = np.full(m * wo, 77, np.int32), np.full(m * wo, 77.0), np.full(m, -5, np.int32), ctypes.c_int32(0)
    assert L.aoclsparse_dcsr2ellthyb(m, base, ctypes.byref(em2), P._ptr(rp), P._ptr(ci), P._ptr(v), None, P._ptr(gm),
                                     P._ptr(gc), P._ptr(gv), wo) == 0
    assert em2.value == emo and np.array_equal(gm[: m - emo], mp) and np.array_equal(gc, hc) and np.array_equal(gv, hv)
    # the reference's own 3x3 (ellmv_tests.cpp:151-204)
    c = kats["ell"][0]
    rp3, ci3, v3 = np.array(c["row_ptr"], np.int32), np.array(c["col_ind"], np.int32), np.array(c["val"])
    gc, gv = np.zeros(6, np.int32), np.zeros(6)
    assert L.aoclsparse_dcsr2ell(3, P.Descr(base=1).h, P._ptr(rp3), P._ptr(ci3), P._ptr(v3), P._ptr(gc), P._ptr(gv), 2) == 0
    assert list(gc) == c["ell_col_ind"] and list(gv) == c["ell_val"]
    # argument checks
    assert L.aoclsparse_csr2ell_width(-1, 0, P._ptr(rp), ctypes.byref(w)) == 3
    assert L.aoclsparse_csr2ell_width(m, 0, None, ctypes.byref(w)) == 2
    assert L.aoclsparse_dcsr2ell(m, d.h, None, P._ptr(ci), P._ptr(v), P._ptr(gc), P._ptr(gv), 3) == 2
    assert L.aoclsparse_dcsr2ell(m, d.h, P._ptr(rp), P._ptr(ci), P._ptr(v), P._ptr(gc), P._ptr(gv), -1) == 3


def test_ellmv_argument_checks():
    """ellmv_tests.cpp:31-150: nullptr, wrong size, not implemented, invalid base, do-nothing."""
    col, val = np.array([-1, 1], np.int32), np.array([0.0, 42.0])
    x, y = np.array([1.0, -2.0, 3.0]), np.array([0.1, 0.2])
    a, b = np.array([2.3]), np.array([11.2])
    d = P.Descr()
    for fn in (L.aoclsparse_dellmv, L.aoclsparse_delltmv):
        args = lambda **k: [k.get("op", P.OP_NONE), P._ptr(a), k.get("m", 2), k.get("n", 3), 1, k.get("val", P._ptr(val)),
                            k.get("col", P._ptr(col)), k.get("w", 1), k.get("d", d.h), k.get("x", P._ptr(x)), P._ptr(b),
                            k.get("y", P._ptr(y))]
        assert fn(*args(val=None)) == 2 and fn(*args(col=None)) == 2 and fn(*args(x=None)) == 2
        assert fn(*args(y=None)) == 2 and fn(*args(d=None)) == 2
        assert fn(*args(m=-1)) == 3 and fn(*args(n=-1)) == 3 and fn(*args(w=-1)) == 3
        assert fn(*args(d=P.Descr(mtype=P.TYPE_SYMMETRIC).h)) == 1 and fn(*args(op=P.OP_TRANSPOSE)) == 1
        bad = P.Descr()
        L.aoclsparse_set_mat_index_base(bad.h, 2)
        assert fn(*args(m=0, w=1)) == 3 and fn(*args(w=0)) == 0 and np.array_equal(y, [0.1, 0.2])
    assert L.aoclsparse_sellthybmv(P.OP_NONE, None, 1, 1, 1, None, None, 1, 1, None, None, None, None, None, d.h, None, None, None) == 1


def _blk_convert(m, n, base, rp, ci, v, rows):
    nnz = len(v)
    brp, bc = np.full(m + 1, -7, np.int32), np.full(max(1, nnz), -7, np.int32)
    bv, mk = np.full(nnz + rows * 8, 77.0), np.full(max(1, nnz) * rows + rows * 8, 0xEE, np.uint8)
    st = L.aoclsparse_csr2blkcsr(m, n, nnz, P._ptr(rp), P._ptr(ci), P._ptr(v), P._ptr(brp), P._ptr(bc), P._ptr(bv),
                                 P._ptr(mk), rows, base)
    return st, brp, bc, bv, mk


@pytest.mark.parametrize("base", [0, 1])
def test_blkcsr_conversion_and_block_size_match_the_oracle(base, kats):
    """aoclsparse_csr2blkcsr / aoclsparse_opt_blksize (conversion/aoclsparse_convert.cpp:36-310) are host integer
    routines: bit-exact against the restatement (pinned on blkcsrmv_tests.cpp's arrays), including windows
    re-anchored at n-8 and row counts that are not a multiple of the block height."""
    import oracle
    from util import banded_rows
    for seed, m, n, per in ((1, 203, 150, lambda r, i: 0 if i % 11 == 3 else 4 + (i * 5) % 17),
                            (2, 61, 19, lambda r, i: 3 + i % 9), (3, 40, 8, lambda r, i: 1 + i % 8),
                            (4, 300, 400, lambda r, i: 12 + i % 30)):
        rp, ci, v = banded_rows(seed, m, n, per, base)
        for rows in (1, 2, 4):
            st, brp, bc, bv, mk = _blk_convert(m, n, base, rp, ci, v, rows)
            so, obrp, obc, obv, omk = oracle.csr2blkcsr(m, n, base, rp, ci, v, rows)
            assert st == so == 0
            nb = len(obc)
            assert np.array_equal(brp, obrp) and np.array_equal(bc[:nb], obc) and np.array_equal(bv[: len(v)], obv)
            assert np.array_equal(mk[: nb * rows], omk)
            assert np.all(bc[nb:] == -7) and np.all(bv[len(v):] == 77.0) and np.all(mk[nb * rows:] == 0xEE)
        tot, oref = ctypes.c_int32(-1), oracle.opt_blksize(m, len(v), base, rp, ci)
        r = L.aoclsparse_opt_blksize(m, len(v), base, P._ptr(rp), P._ptr(ci), ctypes.byref(tot))
        assert r == oref[0] and (r == 0 or tot.value == oref[1])
    # a dense-ish banded matrix makes the heuristic pick a blocked size at all
    rp, ci, v = banded_rows(9, 400, 64, lambda r, i: 40, base)
    tot, oref = ctypes.c_int32(-1), oracle.opt_blksize(400, len(v), base, rp, ci)
    assert L.aoclsparse_opt_blksize(400, len(v), base, P._ptr(rp), P._ptr(ci), ctypes.byref(tot)) == oref[0] != 0
    assert tot.value == oref[1]
    # the reference's own arrays (blkcsrmv_tests.cpp:444-470 == csr2blkcsr of :518-537 with 2x8 blocks)
    d, c = kats["blkcsr"]["direct"], kats["blkcsr"]["csr"][0]
    st, brp, bc, bv, mk = _blk_convert(c["m"], c["n"], 1, np.array(c["row_ptr"], np.int32), np.array(c["col_ind"], np.int32),
                                       np.array(c["val"]), 2)
    assert st == 0 and list(brp) == d["blk_row_ptr"] and list(bc[:3]) == d["blk_col_ind"] and list(mk[:6]) == d["masks"]
    assert list(bv[:14]) == d["val"]


def test_blkcsr_argument_checks():
    """blkcsrmv_tests.cpp:33-300 and :776-900: pointer / size / base / type checks, do-nothing sizes."""
    val, col, ptr = np.array([3.0, 2.0, 1.0]), np.array([1], np.int32), np.array([0, 1, 1], np.int32)
    x, y = np.arange(8.0), np.array([0.1, 0.2])
    a, b, mk = np.array([2.3]), np.array([11.2]), np.array([100, 25], np.uint8)
    d = P.Descr()
    for rows in (1, 2, 4):
        args = lambda **k: [k.get("op", P.OP_NONE), P._ptr(a), k.get("m", 2), k.get("n", 8), k.get("nnz", 3),
                            k.get("mk", P._ptr(mk)), k.get("val", P._ptr(val)), k.get("col", P._ptr(col)),
                            k.get("ptr", P._ptr(ptr)), k.get("d", d.h), k.get("x", P._ptr(x)), P._ptr(b), k.get("y", P._ptr(y)),
                            k.get("rows", rows)]
        fn = L.aoclsparse_dblkcsrmv
        for name in ("mk", "val", "col", "ptr", "x", "y", "d"):
            assert fn(*args(**{name: None})) == 2, name
        assert fn(*args(m=-1)) == 3 and fn(*args(n=-1)) == 3 and fn(*args(n=7)) == 3 and fn(*args(nnz=-1)) == 3
        assert fn(*args(m=0)) == 0 and fn(*args(nnz=0)) == 0 and np.array_equal(y, [0.1, 0.2])
        assert fn(*args(op=P.OP_TRANSPOSE)) == 1 and fn(*args(d=P.Descr(mtype=P.TYPE_TRIANGULAR).h)) == 1
        bad = P.Descr()
        L.aoclsparse_set_mat_index_base(bad.h, 2)
        assert L.aoclsparse_get_mat_index_base(bad.h) == 0  # the setter refuses it ...
        ctypes.cast(bad.h, ctypes.POINTER(ctypes.c_int32))[3] = 2  # ... so write descr->base as blkcsrmv_tests.cpp:413 does
        assert fn(*args(d=bad.h)) == 5
    assert fn(*args(rows=-1)) == 3 and fn(*args(rows=5)) == 3
    tot = ctypes.c_int32(0)
    assert L.aoclsparse_opt_blksize(0, 3, 0, P._ptr(ptr), P._ptr(col), ctypes.byref(tot)) == 0
    assert L.aoclsparse_opt_blksize(-1, 3, 0, P._ptr(ptr), P._ptr(col), ctypes.byref(tot)) == 0
    assert L.aoclsparse_opt_blksize(2, -1, 0, P._ptr(ptr), P._ptr(col), ctypes.byref(tot)) == 0
    assert L.aoclsparse_opt_blksize(2, 3, 0, None, P._ptr(col), ctypes.byref(tot)) == 0
    assert L.aoclsparse_opt_blksize(2, 3, 0, P._ptr(ptr), None, ctypes.byref(tot)) == 0
    assert L.aoclsparse_opt_blksize(2, 3, 0, P._ptr(ptr), P._ptr(col), None) == 0
    out = [np.zeros(8, np.int32), np.zeros(8, np.int32), np.zeros(40), np.zeros(40, np.uint8)]
    cv = lambda **k: [k.get("m", 2), k.get("n", 8), k.get("nnz", 1), k.get("ptr", P._ptr(ptr)), k.get("col", P._ptr(col)),
                      k.get("val", P._ptr(val)), k.get("o0", P._ptr(out[0])), k.get("o1", P._ptr(out[1])),
                      k.get("o2", P._ptr(out[2])), k.get("o3", P._ptr(out[3])), k.get("rows", 2), 0]
    fn = L.aoclsparse_csr2blkcsr
    assert fn(*cv()) == 0
    for name in ("ptr", "col", "val", "o0", "o1", "o2", "o3"):
        assert fn(*cv(**{name: None})) == 2, name
    assert fn(*cv(m=-1)) == 3 and fn(*cv(n=7)) == 3 and fn(*cv(nnz=-1)) == 3 and fn(*cv(rows=3)) == 3 and fn(*cv(rows=0)) == 3


def test_csc_coo_handles_convert_order_and_csr2csc():
    """formats either side of the path (aoclsparse_auxiliary.h:674-1095, aoclsparse_convert.h:494-660): all host
    structure work, int-exact against the oracle's csr2csc restatement and a dense reconstruction."""
    import oracle
    rng = np.random.default_rng(91)
    m, n = 23, 17
    dense = (rng.uniform(size=(m, n)) < 0.3) * rng.uniform(-1, 1, (m, n))
    dense[4, :] = 0.0
    for base in (0, 1):
        rows = [np.flatnonzero(dense[i]) for i in range(m)]
        rp = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int32) + base
        ci = (np.concatenate(rows) + base).astype(np.int32)
        v = np.concatenate([dense[i, r] for i, r in enumerate(rows)])
        nnz = len(v)
        # public csr2csc == the oracle's restatement
        ri, cp, cv = np.zeros(nnz, np.int32), np.zeros(n + 1, np.int32), np.zeros(nnz)
        d = P.Descr(base=base)
        assert L.aoclsparse_dcsr2csc(m, n, nnz, d.h, 1 - base, P._ptr(rp), P._ptr(ci), P._ptr(v), P._ptr(ri), P._ptr(cp), P._ptr(cv)) == 0
        o = oracle.dcsr2csc(m, n, nnz, base, 1 - base, rp, ci, v)
        assert o[0] == 0 and np.array_equal(cp, o[1]) and np.array_equal(ri, o[2]) and np.array_equal(cv, o[3])
        # CSC handle: exports the caller's arrays, behaves as the CSR of the same matrix
        ri, cp, cv = ri - (1 - base) + base, cp - (1 - base) + base, cv.copy()
        h = ctypes.c_void_p()
        assert L.aoclsparse_create_dcsc(ctypes.byref(h), base, m, n, nnz, P._ptr(cp), P._ptr(ri), P._ptr(cv)) == 0
        b_, m_, n_, z_ = ctypes.c_int(), ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
        a1, a2, a3 = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        outs = (ctypes.byref(b_), ctypes.byref(m_), ctypes.byref(n_), ctypes.byref(z_), ctypes.byref(a1), ctypes.byref(a2), ctypes.byref(a3))
        assert L.aoclsparse_export_dcsc(h, *outs) == 0
        assert (a1.value, a2.value, a3.value) == (cp.ctypes.data, ri.ctypes.data, cv.ctypes.data) and (m_.value, n_.value) == (m, n)
        assert L.aoclsparse_export_dcsr(h, *outs) == 0 and (b_.value, m_.value, n_.value, z_.value) == (base, m, n, nnz)
        g_rp = np.ctypeslib.as_array(ctypes.cast(a1, ctypes.POINTER(ctypes.c_int32)), (m + 1,))
        g_ci = np.ctypeslib.as_array(ctypes.cast(a2, ctypes.POINTER(ctypes.c_int32)), (nnz,))
        g_v = np.ctypeslib.as_array(ctypes.cast(a3, ctypes.POINTER(ctypes.c_double)), (nnz,))
        assert np.array_equal(g_rp, rp) and np.array_equal(g_ci, ci) and np.array_equal(g_v, v)
        assert L.aoclsparse_dset_value(h, 2 + base, int(ci[rp[2] - base]), 9.5) == 0  # written to CSC and CSR
        assert 9.5 in cv and g_v[rp[2] - base] == 9.5
        assert L.aoclsparse_export_dcoo(h, *outs) == 5
        L.aoclsparse_destroy(ctypes.byref(h))
        # COO handle in shuffled order -> convert_csr (op none / transpose)
        perm = rng.permutation(nnz)
        cr = (np.repeat(np.arange(m), np.diff(rp)) + base).astype(np.int32)[perm]
        cc, cval = ci[perm].copy(), v[perm].copy()
        assert L.aoclsparse_create_dcoo(ctypes.byref(h), base, m, n, nnz, P._ptr(cr), P._ptr(cc), P._ptr(cval)) == 0
        assert L.aoclsparse_export_dcoo(h, *outs) == 0 and a1.value == cr.ctypes.data
        for op, mm, nn, ref in ((P.OP_NONE, m, n, dense), (P.OP_TRANSPOSE, n, m, dense.T)):
            c = ctypes.c_void_p()
            assert L.aoclsparse_convert_csr(h, op, ctypes.byref(c)) == 0
            assert L.aoclsparse_order_mat(c) == 0  # rows hold the COO order: sort them
            assert L.aoclsparse_export_dcsr(c, *outs) == 0 and (m_.value, n_.value, z_.value, b_.value) == (mm, nn, nnz, base)
            q_rp = np.ctypeslib.as_array(ctypes.cast(a1, ctypes.POINTER(ctypes.c_int32)), (mm + 1,))
            q_ci = np.ctypeslib.as_array(ctypes.cast(a2, ctypes.POINTER(ctypes.c_int32)), (nnz,))
            q_v = np.ctypeslib.as_array(ctypes.cast(a3, ctypes.POINTER(ctypes.c_double)), (nnz,))
            rebuilt = np.zeros((mm, nn))
            for i in range(mm):
                seg = slice(q_rp[i] - base, q_rp[i + 1] - base)
                assert np.all(np.diff(q_ci[seg]) > 0)
                rebuilt[i, q_ci[seg] - base] = q_v[seg]
            assert np.array_equal(rebuilt, ref)
            L.aoclsparse_destroy(ctypes.byref(c))
        x, y, one, zero = np.ones(n), np.zeros(m), np.array([1.0]), np.array([0.0])
        assert L.aoclsparse_dmv(P.OP_NONE, P._ptr(one), h, d.h, P._ptr(x), P._ptr(zero), P._ptr(y)) == 1  # COO: not_implemented
        bad = cr.copy()
        bad[0] = m + base
        g = ctypes.c_void_p()
        assert L.aoclsparse_create_dcoo(ctypes.byref(g), base, m, n, nnz, P._ptr(bad), P._ptr(cc), P._ptr(cval)) == 6
        L.aoclsparse_destroy(ctypes.byref(h))
    # order_mat on an unsorted CSR handle sorts the caller's arrays in place
    rp = np.array([0, 2, 3, 4, 7, 8], np.int32)
    ci = np.array([3, 0, 1, 2, 3, 1, 4, 4], np.int32)
    v = np.array([2.0, 1, 3, 4, 6, 5, 7, 8])
    A = P.Matrix(0, 5, 5, rp, ci, v)
    assert L.aoclsparse_order_mat(A.h) == 0
    assert list(ci) == [0, 3, 1, 2, 1, 3, 4, 4] and list(v) == [1, 2, 3, 4, 5, 6, 7, 8]
    assert L.aoclsparse_order_mat(None) == 2


@pytest.mark.parametrize("prec", ["c", "z"])
def test_complex_csc_coo_convert_order_and_csr2csc(prec):
    """complex twins of the format routines (aoclsparse_auxiliary.h:438-560, aoclsparse_convert.h:528-560): host structure
    work checked against a dense reconstruction; aoclsparse_convert_csr with op = H conjugates."""
    ct, rt, exp_csr = (np.complex64, np.float32, L.aoclsparse_export_ccsr) if prec == "c" else (np.complex128, np.float64, L.aoclsparse_export_zcsr)
    fn = lambda stem: getattr(L, "aoclsparse_" + stem.replace("?", prec))
    rng = np.random.default_rng(17)
    m, n = 19, 13
    dense = ((rng.uniform(size=(m, n)) < 0.35) * (rng.uniform(-1, 1, (m, n)) + 1j * rng.uniform(-1, 1, (m, n)))).astype(ct)
    dense[3, :] = 0
    for base in (0, 1):
        rows = [np.flatnonzero(dense[i]) for i in range(m)]
        rp = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int32) + base
        ci = (np.concatenate(rows) + base).astype(np.int32)
        v = np.concatenate([dense[i, r] for i, r in enumerate(rows)]).astype(ct)
        nnz = len(v)
        d = P.Descr(base=base)
        ri, cp, cv = np.zeros(nnz, np.int32), np.zeros(n + 1, np.int32), np.zeros(nnz, ct)
        assert fn("?csr2csc")(m, n, nnz, d.h, base, P._ptr(rp), P._ptr(ci), P._ptr(v), P._ptr(ri), P._ptr(cp), P._ptr(cv)) == 0
        back = np.zeros((m, n), ct)
        for j in range(n):
            seg = slice(cp[j] - base, cp[j + 1] - base)
            assert np.all(np.diff(ri[seg]) > 0)
            back[ri[seg] - base, j] = cv[seg]
        assert np.array_equal(back, dense)
        b_, m_, n_, z_ = ctypes.c_int(), ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
        a1, a2, a3 = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        outs = (ctypes.byref(b_), ctypes.byref(m_), ctypes.byref(n_), ctypes.byref(z_), ctypes.byref(a1), ctypes.byref(a2), ctypes.byref(a3))

        def csr_to_dense(h, mm, nn):
            assert exp_csr(h, *outs) == 0 and (m_.value, n_.value, z_.value, b_.value) == (mm, nn, nnz, base)
            q_rp = np.ctypeslib.as_array(ctypes.cast(a1, ctypes.POINTER(ctypes.c_int32)), (mm + 1,))
            q_ci = np.ctypeslib.as_array(ctypes.cast(a2, ctypes.POINTER(ctypes.c_int32)), (nnz,))
            q_v = np.ctypeslib.as_array(ctypes.cast(a3, ctypes.POINTER(ctypes.c_float if prec == "c" else ctypes.c_double)), (2 * nnz,))
            q_v = q_v.view(ct)
            out = np.zeros((mm, nn), ct)
            for i in range(mm):
                seg = slice(q_rp[i] - base, q_rp[i + 1] - base)
                assert np.all(np.diff(q_ci[seg]) > 0)
                out[i, q_ci[seg] - base] = q_v[seg]
            return out

        # CSC handle: its CSR is the same matrix; export_?csc hands back the caller's arrays
        h = ctypes.c_void_p()
        assert fn("create_?csc")(ctypes.byref(h), base, m, n, nnz, P._ptr(cp), P._ptr(ri), P._ptr(cv)) == 0
        assert fn("export_?csc")(h, *outs) == 0 and (a1.value, a2.value, a3.value) == (cp.ctypes.data, ri.ctypes.data, cv.ctypes.data)
        assert np.array_equal(csr_to_dense(h, m, n), dense)
        wrong = L.aoclsparse_export_dcsc if prec == "z" else L.aoclsparse_export_zcsc
        assert wrong(h, *outs) == 9  # aoclsparse_status_wrong_type
        L.aoclsparse_destroy(ctypes.byref(h))
        # COO handle in shuffled order -> convert_csr for the three operations
        perm = rng.permutation(nnz)
        cr = (np.repeat(np.arange(m), np.diff(rp)) + base).astype(np.int32)[perm]
        cc, cval = ci[perm].copy(), v[perm].copy()
        assert fn("create_?coo")(ctypes.byref(h), base, m, n, nnz, P._ptr(cr), P._ptr(cc), P._ptr(cval)) == 0
        assert fn("export_?coo")(h, *outs) == 0 and a3.value == cval.ctypes.data
        for op, mm, nn, ref in ((P.OP_NONE, m, n, dense), (P.OP_TRANSPOSE, n, m, dense.T), (P.OP_CONJ_TRANSPOSE, n, m, dense.conj().T)):
            c = ctypes.c_void_p()
            assert L.aoclsparse_convert_csr(h, op, ctypes.byref(c)) == 0
            assert L.aoclsparse_order_mat(c) == 0
            assert np.array_equal(csr_to_dense(c, mm, nn), ref)
            L.aoclsparse_destroy(ctypes.byref(c))
        L.aoclsparse_destroy(ctypes.byref(h))


def test_csrsv_argument_checks():
    """level2/aoclsparse_csrsv.hpp:38-75, in its order (all before any device work)."""
    rp, ci, v = np.array([0, 1, 3], np.int32), np.array([0, 0, 1], np.int32), np.array([2.0, 1.0, 4.0])
    x, y, a = np.array([1.0, 2.0]), np.zeros(2), np.array([1.0])
    d = P.Descr()
    args = lambda **k: [k.get("op", P.OP_NONE), k.get("a", P._ptr(a)), k.get("m", 2), k.get("val", P._ptr(v)), k.get("col", P._ptr(ci)),
                        k.get("ptr", P._ptr(rp)), k.get("d", d.h), k.get("x", P._ptr(x)), k.get("y", P._ptr(y))]
    fn = L.aoclsparse_dcsrsv
    for name in ("a", "val", "col", "ptr", "d", "x", "y"):
        assert fn(*args(**{name: None})) == 2, name
    assert fn(*args(d=P.Descr(base=1).h)) == 1 and fn(*args(d=P.Descr(mtype=P.TYPE_TRIANGULAR).h)) == 1
    assert fn(*args(op=P.OP_TRANSPOSE)) == 1 and fn(*args(m=-1)) == 3 and fn(*args(m=0)) == 0
    assert L.aoclsparse_scsrsv(P.OP_NONE, None, 2, None, None, None, None, None, None) == 2


def test_dense_result_and_add_argument_checks():
    """sp2md.hpp:192-276, spmmd.cpp:47-55, convert.hpp:669-727, csradd.hpp:289-312: every check that precedes the
    computation, in the reference's order; an empty sum still yields a valid empty handle (csradd.hpp:163-177)."""
    rp, ci, v = np.array([0, 1, 3], np.int32), np.array([0, 0, 1], np.int32), np.array([2.0, 1.0, 4.0])
    A, Af = P.Matrix(0, 2, 2, rp, ci, v), P.Matrix(0, 2, 2, rp, ci, v.astype(np.float32))
    W = P.Matrix(0, 2, 3, rp, np.array([0, 0, 2], np.int32), v)
    d, ds, d1 = P.Descr(), P.Descr(mtype=P.TYPE_SYMMETRIC), P.Descr(base=1)
    C = np.zeros(6)
    f = L.aoclsparse_dsp2md
    N, T, ROW, COL = P.OP_NONE, P.OP_TRANSPOSE, P.ORDER_ROW, P.ORDER_COLUMN
    assert f(N, None, A.h, N, d.h, A.h, 1.0, 0.0, P._ptr(C), ROW, 2) == 2
    assert f(N, ds.h, A.h, N, d.h, A.h, 1.0, 0.0, P._ptr(C), ROW, 2) == 1
    assert f(N, d.h, A.h, N, d.h, A.h, 1.0, 0.0, P._ptr(C), 7, 2) == 5
    assert f(N, d.h, None, N, d.h, A.h, 1.0, 0.0, P._ptr(C), ROW, 2) == 2
    assert f(N, d.h, A.h, N, d.h, A.h, 1.0, 0.0, None, ROW, 2) == 2
    assert f(N, d.h, A.h, N, d.h, Af.h, 1.0, 0.0, P._ptr(C), ROW, 2) == 9
    assert f(N, d.h, W.h, N, d.h, A.h, 1.0, 0.0, P._ptr(C), ROW, 3) == 3  # 2x3 * 2x2
    assert f(T, d.h, W.h, N, d.h, A.h, 1.0, 0.0, P._ptr(C), ROW, 1) == 3  # ldc < n
    assert f(T, d.h, W.h, N, d.h, A.h, 1.0, 0.0, P._ptr(C), COL, 2) == 3  # ldc < rows of C (3)
    assert f(N, d1.h, A.h, N, d.h, A.h, 1.0, 0.0, P._ptr(C), ROW, 2) == 5  # descriptor base != matrix base
    assert L.aoclsparse_dspmmd(N, None, A.h, ROW, P._ptr(C), 2) == 2
    assert L.aoclsparse_dspmmd(N, A.h, Af.h, ROW, P._ptr(C), 2) == 9
    assert L.aoclsparse_sspmmd(N, Af.h, Af.h, 9, P._ptr(C), 2) == 5
    g = L.aoclsparse_dcsr2dense
    D = np.zeros(4)
    assert g(2, 2, None, P._ptr(v), P._ptr(rp), P._ptr(ci), P._ptr(D), 2, ROW) == 2
    assert g(2, 2, P.Descr(mtype=P.TYPE_TRIANGULAR).h, P._ptr(v), P._ptr(rp), P._ptr(ci), P._ptr(D), 2, COL) == 1
    assert g(2, 2, P.Descr(mtype=P.TYPE_HERMITIAN).h, P._ptr(v), P._ptr(rp), P._ptr(ci), P._ptr(D), 2, COL) == 1
    assert g(-1, 2, d.h, P._ptr(v), P._ptr(rp), P._ptr(ci), P._ptr(D), 2, ROW) == 3
    assert g(0, 2, d.h, None, None, None, None, 2, ROW) == 0
    for k in range(4):
        a = [P._ptr(v), P._ptr(rp), P._ptr(ci), P._ptr(D)]
        a[k] = None
        assert g(2, 2, d.h, a[0], a[1], a[2], a[3], 2, ROW) == 2
    assert g(2, 2, d.h, P._ptr(v), P._ptr(rp), P._ptr(ci), P._ptr(D), 2 ** 31 - 1, ROW) == 3
    h = L.aoclsparse_dadd
    out = ctypes.c_void_p()
    assert h(N, None, 1.0, A.h, ctypes.byref(out)) == 2 and h(N, A.h, 1.0, A.h, None) == 2
    assert h(N, A.h, 1.0, Af.h, ctypes.byref(out)) == 9
    assert h(N, A.h, 1.0, W.h, ctypes.byref(out)) == 3 and h(T, W.h, 1.0, W.h, ctypes.byref(out)) == 3
    e0 = np.array([1, 1, 1], np.int32)
    E = P.Matrix(1, 2, 2, e0, np.zeros(1, np.int32), np.zeros(1))
    assert h(N, E.h, 1.0, E.h, ctypes.byref(out)) == 0 and out.value
    base, m, n, nnz = ctypes.c_int(), ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
    p0, p1, p2 = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
    assert L.aoclsparse_export_dcsr(out, ctypes.byref(base), ctypes.byref(m), ctypes.byref(n), ctypes.byref(nnz),
                                    ctypes.byref(p0), ctypes.byref(p1), ctypes.byref(p2)) == 0
    assert (base.value, m.value, n.value, nnz.value) == (1, 2, 2, 0)
    assert list(np.ctypeslib.as_array(ctypes.cast(p0, ctypes.POINTER(ctypes.c_int32)), (3,))) == [1, 1, 1]
    assert L.aoclsparse_destroy(ctypes.byref(out)) == 0


def test_level1_argument_checks():
    """Every check of the level-1 dispatchers that precedes the computation, in the reference's order
    (axpyi.hpp:70-85, dot.hpp:74-86, gthr.hpp:73-104, sctr.hpp:65-88, roti.hpp:70-83)."""
    x, y, ix = np.array([1.0, 2.0]), np.zeros(4), np.array([0, 3], np.int32)
    X, Y, IX = P._ptr(x), P._ptr(y), P._ptr(ix)
    assert L.aoclsparse_daxpyi(2, 1.0, None, IX, Y) == 2 and L.aoclsparse_daxpyi(-1, 1.0, None, IX, Y) == 2
    assert L.aoclsparse_daxpyi(0, 1.0, X, IX, Y) == 0 and L.aoclsparse_daxpyi(-1, 1.0, X, IX, Y) == 3
    assert L.aoclsparse_daxpyi_kid(2, 1.0, X, IX, Y, 4) == 14
    a = np.array([1.0 + 0j])
    assert L.aoclsparse_zaxpyi(2, None, X, IX, Y) == 2 and L.aoclsparse_zaxpyi(0, P._ptr(a), X, IX, Y) == 0
    dot = np.array([7.0 + 7.0j])
    assert L.aoclsparse_zdotci(2, X, IX, Y, None) == 2
    assert L.aoclsparse_zdotui(0, X, IX, Y, P._ptr(dot)) == 3 and dot[0] == 0  # dot.hpp:78-82: cleared
    assert L.aoclsparse_zdotci(2, None, IX, Y, P._ptr(dot)) == 2 and L.aoclsparse_cdotui_kid(2, X, IX, Y, P._ptr(dot), 9) == 14
    assert L.aoclsparse_ddoti(0, X, IX, Y) == 0.0 and L.aoclsparse_sdoti(-3, None, None, None) == 0.0
    for f in (L.aoclsparse_dgthr, L.aoclsparse_dgthrz, L.aoclsparse_sgthr, L.aoclsparse_zgthrz):
        assert f(-1, Y, X, IX) == 3 and f(0, None, None, None) == 0
        assert f(2, None, X, IX) == 2 and f(2, Y, None, IX) == 2 and f(2, Y, X, None) == 2
    assert L.aoclsparse_dgthrs(2, Y, X, -1) == 3 and L.aoclsparse_dgthrs(2, None, X, 1) == 2
    assert L.aoclsparse_dgthr_kid(2, Y, X, IX, 4) == 14
    for f in (L.aoclsparse_dsctr, L.aoclsparse_csctr):
        assert f(2, None, IX, Y) == 2 and f(2, X, IX, None) == 2 and f(0, X, IX, Y) == 0
        assert f(-1, X, IX, Y) == 3 and f(2, X, None, Y) == 2
    assert L.aoclsparse_dsctrs(2, X, 0, Y) == 3 and L.aoclsparse_dsctrs(2, X, -2, Y) == 3 and L.aoclsparse_dsctrs(0, X, 0, Y) == 0
    assert L.aoclsparse_ssctrs_kid(2, X, 1, Y, 5) == 14
    assert L.aoclsparse_droti(2, None, IX, Y, 1.0, 0.0) == 2 and L.aoclsparse_droti(0, X, IX, Y, 1.0, 0.0) == 0
    assert L.aoclsparse_droti(-2, X, IX, Y, 1.0, 0.0) == 3 and L.aoclsparse_sroti_kid(2, X, IX, Y, 1.0, 0.0, 4) == 14


@pytest.mark.parametrize("base", [0, 1])
def test_dia_bsr_conversions_match_the_oracle(base, kats):
    """aoclsparse_csr2dia_ndiag / ?csr2dia / csr2bsr_nnz / ?csr2bsr are host routines: bit-exact against the oracle
    (and through it the reference's vectors), plus their argument checks (convert.cpp:518-531, :605-637,
    convert.hpp:301-334, :401-440) and those of ?diamv / ?bsrmv that precede any device work."""
    m, n = 203, 167
    rp, ci, v = random_csr(95, m, n, lambda r, i: r.integers(0, 9), base=base, sort=False)
    d = P.Descr(base=base)
    nd = ctypes.c_int32(-1)
    assert L.aoclsparse_csr2dia_ndiag(m, n, d.h, len(v), P._ptr(rp), P._ptr(ci), ctypes.byref(nd)) == 0
    ond, ooff, odv = oracle.csr2dia(m, n, base, rp, ci, v)
    off, dv = np.zeros(nd.value, np.int32), np.zeros(nd.value * m)
    assert nd.value == ond
    assert L.aoclsparse_dcsr2dia(m, n, d.h, P._ptr(rp), P._ptr(ci), P._ptr(v), nd.value, P._ptr(off), P._ptr(dv)) == 0
    assert np.array_equal(off, ooff) and np.array_equal(dv, odv)
    vf, dvf = v.astype(np.float32), np.zeros(nd.value * m, np.float32)
    assert L.aoclsparse_scsr2dia(m, n, d.h, P._ptr(rp), P._ptr(ci), P._ptr(vf), nd.value, P._ptr(off), P._ptr(dvf)) == 0
    assert np.array_equal(dvf, odv.astype(np.float32))
    for dim in (1, 2, 5, 8):
        for order, rowmajor in ((P.ORDER_ROW, True), (P.ORDER_COLUMN, False)):
            mb = (m + dim - 1) // dim
            bp, nnzb = np.zeros(mb + 1, np.int32), ctypes.c_int32(-1)
            assert L.aoclsparse_csr2bsr_nnz(m, n, d.h, P._ptr(rp), P._ptr(ci), dim, P._ptr(bp), ctypes.byref(nnzb)) == 0
            obp, obi, obv = oracle.csr2bsr(m, n, base, rp, ci, v, dim, rowmajor)
            assert nnzb.value == len(obi) and np.array_equal(bp, obp)
            bi, bv = np.zeros(nnzb.value, np.int32), np.zeros(nnzb.value * dim * dim)
            assert L.aoclsparse_dcsr2bsr(m, n, d.h, order, P._ptr(v), P._ptr(rp), P._ptr(ci), dim, P._ptr(bv), P._ptr(bp), P._ptr(bi)) == 0
            assert np.array_equal(bi, obi) and np.array_equal(bv, obv)
            vz, bz = (v + 2j * v).astype(np.complex128), np.zeros(nnzb.value * dim * dim, np.complex128)
            assert L.aoclsparse_zcsr2bsr(m, n, d.h, order, P._ptr(vz), P._ptr(rp), P._ptr(ci), dim, P._ptr(bz), P._ptr(bp), P._ptr(bi)) == 0
            assert np.array_equal(bz, obv + 2j * obv)
    # argument checks
    assert L.aoclsparse_csr2dia_ndiag(-1, n, d.h, 1, P._ptr(rp), P._ptr(ci), ctypes.byref(nd)) == 3
    assert L.aoclsparse_csr2dia_ndiag(m, n, d.h, 1, P._ptr(rp), P._ptr(ci), None) == 2
    assert L.aoclsparse_csr2dia_ndiag(m, n, d.h, 1, None, P._ptr(ci), ctypes.byref(nd)) == 2
    assert L.aoclsparse_dcsr2dia(m, n, d.h, P._ptr(rp), P._ptr(ci), P._ptr(v), -1, P._ptr(off), P._ptr(dv)) == 3
    assert L.aoclsparse_dcsr2dia(m, n, d.h, P._ptr(rp), P._ptr(ci), P._ptr(v), 0, None, None) == 0
    assert L.aoclsparse_dcsr2dia(m, n, d.h, P._ptr(rp), P._ptr(ci), None, 3, P._ptr(off), P._ptr(dv)) == 2
    bp, nnzb = np.zeros(m + 1, np.int32), ctypes.c_int32(-1)
    assert L.aoclsparse_csr2bsr_nnz(m, n, d.h, P._ptr(rp), P._ptr(ci), 0, P._ptr(bp), ctypes.byref(nnzb)) == 3
    assert L.aoclsparse_csr2bsr_nnz(m, n, d.h, P._ptr(rp), None, 2, P._ptr(bp), ctypes.byref(nnzb)) == 2
    assert L.aoclsparse_csr2bsr_nnz(0, n, d.h, P._ptr(rp), P._ptr(ci), 2, P._ptr(bp), ctypes.byref(nnzb)) == 0 and nnzb.value == 0
    assert L.aoclsparse_dcsr2bsr(m, n, d.h, P.ORDER_ROW, P._ptr(v), P._ptr(rp), P._ptr(ci), 0, P._ptr(v), P._ptr(bp), P._ptr(bp)) == 5
    assert L.aoclsparse_dcsr2bsr(m, n, d.h, P.ORDER_ROW, None, P._ptr(rp), P._ptr(ci), 2, P._ptr(v), P._ptr(bp), P._ptr(bp)) == 2
    a, b, x, y = ctypes.c_double(1.0), ctypes.c_double(0.0), np.zeros(n), np.zeros(m)
    A, B = ctypes.byref(a), ctypes.byref(b)
    f = L.aoclsparse_ddiamv
    assert f(P.OP_NONE, None, m, n, 0, P._ptr(dv), P._ptr(off), 1, d.h, P._ptr(x), B, P._ptr(y)) == 2
    assert f(P.OP_NONE, A, m, n, 0, P._ptr(dv), P._ptr(off), 1, None, P._ptr(x), B, P._ptr(y)) == 2
    assert f(P.OP_TRANSPOSE, A, m, n, 0, P._ptr(dv), P._ptr(off), 1, d.h, P._ptr(x), B, P._ptr(y)) == 1
    assert f(P.OP_NONE, A, m, n, 0, P._ptr(dv), P._ptr(off), 1, P.Descr(base=base, mtype=P.TYPE_SYMMETRIC).h, P._ptr(x), B, P._ptr(y)) == 1
    assert f(P.OP_NONE, A, m, n, 0, P._ptr(dv), P._ptr(off), -1, d.h, P._ptr(x), B, P._ptr(y)) == 3
    assert f(P.OP_NONE, A, 0, n, 0, P._ptr(dv), P._ptr(off), 1, d.h, P._ptr(x), B, P._ptr(y)) == 0
    assert L.aoclsparse_ddiamv_kid(P.OP_NONE, A, m, n, 0, P._ptr(dv), P._ptr(off), 1, d.h, P._ptr(x), B, P._ptr(y), 0, 4) == 14
    g = L.aoclsparse_dbsrmv
    assert g(P.OP_NONE, None, 2, 2, 2, P._ptr(v), P._ptr(ci), P._ptr(rp), d.h, P._ptr(x), B, P._ptr(y)) == 2
    assert g(P.OP_NONE, A, 2, 2, 2, P._ptr(v), P._ptr(ci), P._ptr(rp), None, P._ptr(x), B, P._ptr(y)) == 2
    assert g(P.OP_TRANSPOSE, A, 2, 2, 2, P._ptr(v), P._ptr(ci), P._ptr(rp), d.h, P._ptr(x), B, P._ptr(y)) == 1
    assert g(P.OP_NONE, A, 2, 2, 0, P._ptr(v), P._ptr(ci), P._ptr(rp), d.h, P._ptr(x), B, P._ptr(y)) == 3
    assert g(P.OP_NONE, A, 0, 2, 2, None, None, None, d.h, None, B, None) == 0
    assert g(P.OP_NONE, A, 2, 2, 2, None, P._ptr(ci), P._ptr(rp), d.h, P._ptr(x), B, P._ptr(y)) == 2


def test_sorv_argument_checks():
    """solvers/aoclsparse_sorv.hpp:126-210, in its order; everything up to the diagonal check happens on the host."""
    rp, ci, v = np.array([0, 2, 3], np.int32), np.array([0, 1, 1], np.int32), np.array([2.0, 1.0, 4.0])
    A, d, x, b = P.Matrix(0, 2, 2, rp, ci, v), P.Descr(), np.zeros(2), np.ones(2)
    f = L.aoclsparse_dsorv
    assert f(0, d.h, None, 1.0, 1.0, P._ptr(x), P._ptr(b)) == 2 and f(0, None, A.h, 1.0, 1.0, P._ptr(x), P._ptr(b)) == 2
    assert f(0, d.h, A.h, 1.0, 1.0, None, P._ptr(b)) == 2 and f(0, d.h, A.h, 1.0, 1.0, P._ptr(x), None) == 2
    assert f(0, P.Descr(base=1).h, A.h, 1.0, 1.0, P._ptr(x), P._ptr(b)) == 5
    W = P.Matrix(0, 2, 3, rp, ci, v)
    assert f(0, d.h, W.h, 1.0, 1.0, P._ptr(x), P._ptr(b)) == 3
    assert f(0, P.Descr(mtype=P.TYPE_SYMMETRIC).h, A.h, 1.0, 1.0, P._ptr(x), P._ptr(b)) == 1
    assert L.aoclsparse_ssorv(0, d.h, A.h, 1.0, 1.0, P._ptr(x), P._ptr(b)) == 9
    assert f(1, d.h, A.h, 1.0, 1.0, P._ptr(x), P._ptr(b)) == 1 and f(2, d.h, A.h, 1.0, 1.0, P._ptr(x), P._ptr(b)) == 1
    assert f(7, d.h, A.h, 1.0, 1.0, P._ptr(x), P._ptr(b)) == 5
    Z = P.Matrix(0, 2, 2, rp, np.array([0, 1, 0], np.int32), v)  # second row without a diagonal
    assert f(0, d.h, Z.h, 1.0, 1.0, P._ptr(x), P._ptr(b)) == 5
    assert L.aoclsparse_zsorv(0, d.h, A.h, P.CDouble(1, 0), P.CDouble(1, 0), P._ptr(x), P._ptr(b)) == 1
