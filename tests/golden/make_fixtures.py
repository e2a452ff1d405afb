#!/usr/bin/env python3
"""Generate tests/golden/reference_kats.json: the known-answer vectors the reference's own
unit tests hold for the CSR SpMV / clean-CSR / TRSV / csrmm path.

The small cases are re-typed here as data (matrix, right-hand side, expected result), each with
the reference file:line it comes from.  The N25 TRSV system (565 non-zeros, 16 right-hand
sides) is extracted numerically from the reference's fixture database when /root/reference is
present (this container only); the committed JSON is what the tests read, so nothing under
tests/ touches /root/reference at run time.

Run:  python tests/golden/make_fixtures.py        (rewrites reference_kats.json)
"""
import json
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/tests/unit_tests"

out = {"_about": "known-answer vectors re-typed/extracted from amd/aocl-sparse v5.3.2 unit tests; "
                 "data only (inputs + expected outputs)"}

# ------------------------------------------------------------------------------------------
# SpMV: tests/unit_tests/csrmv_tests.cpp:185-220 (base-1 KAT), tests/examples/sample_spmv_c.c:50-59
# ------------------------------------------------------------------------------------------
out["csrmv"] = [
    dict(name="N5_base0", src="tests/examples/sample_spmv_c.c:50-59", base=0, m=5, n=5,
         row_ptr=[0, 2, 3, 4, 7, 8], col_ind=[0, 3, 1, 2, 1, 3, 4, 4], val=[1, 2, 3, 4, 5, 6, 7, 8],
         x=[1, 2, 3, 4, 5], alpha=1.0, beta=0.0, y0=[0, 0, 0, 0, 0], y_gold=[9, 6, 12, 69, 40]),
    dict(name="N5_base1", src="tests/unit_tests/csrmv_tests.cpp:185-220", base=1, m=5, n=5,
         row_ptr=[1, 3, 4, 5, 8, 9], col_ind=[1, 4, 2, 3, 2, 4, 5, 5], val=[1, 2, 3, 4, 5, 6, 7, 8],
         x=[1, 2, 3, 4, 5], alpha=1.0, beta=0.0, y0=[0, 0, 0, 0, 0], y_gold=[9, 6, 12, 69, 40]),
]

# symmetric raw csrmv: tests/unit_tests/csrmv_tests.cpp:288-352.  The test compares against its own
# dense-free reference (ref_csrmvsym); the matrix and x are small integers, so y_gold below is the exact
# symmetric product (L + D + L^T) x, computed here in integer arithmetic.
_sr = [0, 1, 2, 5, 6, 8, 11, 15, 18]
_sc = [0, 1, 0, 1, 2, 3, 1, 4, 0, 4, 5, 0, 3, 4, 6, 2, 5, 7]
_sv = [19, 10, 1, 8, 11, 13, 2, 11, 2, 1, 9, 7, 9, 5, 12, 5, 5, 9]
_sx = [1, 2, 3, 4, 5, 6, 7, 8]
_sy = [0] * 8
for _i in range(8):
    for _p in range(_sr[_i], _sr[_i + 1]):
        _sy[_i] += _sv[_p] * _sx[_sc[_p]]
        if _sc[_p] != _i:
            _sy[_sc[_p]] += _sv[_p] * _sx[_i]
out["csrmv_sym"] = [dict(name="S8_lower", src="tests/unit_tests/csrmv_tests.cpp:288-352", base=0, m=8,
                         fill="lower", row_ptr=_sr, col_ind=_sc, val=_sv, x=_sx, alpha=1.0, beta=0.0,
                         y_gold=_sy)]

# ------------------------------------------------------------------------------------------
# clean CSR after optimize: inputs tests/unit_tests/common_data_utils.h:609-760,
# expected tests/unit_tests/hint_tests.cpp:75-170 (all base 0, double)
# ------------------------------------------------------------------------------------------
N10_in_col = [9, 4, 6, 3, 8, 6, 0, 6, 4, 6, 7, 1, 2, 9, 3, 8, 5, 0, 6, 2, 1,
              5, 3, 8, 3, 8, 5, 1, 4, 8, 5, 9, 1, 4, 8, 5, 4, 6, 6, 2, 3, 7]
N10_in_val = [5.91, 5.95, 7.95, 0.83, 5.48, 6.75, 0.01, 4.78, 9.20, 3.40, 2.26, 3.01, 8.34, 6.82,
              7.40, 1.12, 3.31, 4.96, 2.66, 1.77, 5.28, 8.95, 3.09, 2.37, 4.48, 2.92, 1.46, 6.17,
              8.77, 9.96, 7.19, 9.61, 6.48, 4.95, 6.76, 8.87, 5.07, 3.58, 2.09, 8.66, 6.77, 3.69]
N10_out_col = [0, 4, 6, 9, 1, 3, 6, 8, 0, 2, 6, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 1, 2, 3, 4,
               5, 6, 8, 3, 5, 8, 1, 4, 5, 6, 8, 1, 4, 5, 7, 8, 9, 4, 6, 8, 2, 3, 6, 7, 9]
N10_out_val = [0, 5.95, 7.95, 5.91, 0, 0.83, 6.75, 5.48, 0.01, 0, 4.78, 4.96, 3.01,
               8.34, 7.4, 9.2, 3.31, 3.4, 2.26, 1.12, 6.82, 5.28, 1.77, 3.09, 0, 8.95,
               2.66, 2.37, 4.48, 1.46, 2.92, 6.17, 8.77, 7.19, 0, 9.96, 6.48, 4.95, 8.87,
               0, 6.76, 9.61, 5.07, 3.58, 0, 8.66, 6.77, 2.09, 3.69, 0]
out["clean_csr"] = [
    dict(name="N5_full_sorted", m=5, n=5,
         row_ptr=[0, 2, 3, 4, 7, 8], col_ind=[0, 3, 1, 2, 1, 3, 4, 4], val=[1, 2, 3, 4, 5, 6, 7, 8],
         exp=dict(icrow=[0, 2, 3, 4, 7, 8], icol=[0, 3, 1, 2, 1, 3, 4, 4],
                  aval=[1, 2, 3, 4, 5, 6, 7, 8], idiag=[0, 2, 3, 5, 7], iurow=[1, 3, 4, 6, 8],
                  is_internal=False)),
    dict(name="N5_full_unsorted", m=5, n=5,
         row_ptr=[0, 2, 3, 4, 7, 8], col_ind=[3, 0, 1, 2, 3, 1, 4, 4], val=[2, 1, 3, 4, 6, 5, 7, 8],
         exp=dict(icrow=[0, 2, 3, 4, 7, 8], icol=[0, 3, 1, 2, 1, 3, 4, 4],
                  aval=[1, 2, 3, 4, 5, 6, 7, 8], idiag=[0, 2, 3, 5, 7], iurow=[1, 3, 4, 6, 8],
                  is_internal=True)),
    dict(name="N59_partial_sort", m=5, n=5,
         row_ptr=[0, 2, 3, 4, 8, 9], col_ind=[0, 3, 1, 2, 2, 1, 3, 4, 4],
         val=[1, 2, 3, 4, 9, 5, 6, 7, 8],
         exp=dict(icrow=[0, 2, 3, 4, 8, 9], icol=[0, 3, 1, 2, 2, 1, 3, 4, 4],
                  aval=[1, 2, 3, 4, 9, 5, 6, 7, 8], idiag=[0, 2, 3, 6, 8], iurow=[1, 3, 4, 7, 9],
                  is_internal=False)),
    dict(name="N5_1_hole", m=5, n=5,
         row_ptr=[0, 2, 3, 4, 6, 7], col_ind=[3, 0, 1, 2, 1, 4, 4], val=[2, 1, 3, 4, 5, 7, 8],
         exp=dict(icrow=[0, 2, 3, 4, 7, 8], icol=[0, 3, 1, 2, 1, 3, 4, 4],
                  aval=[1, 2, 3, 4, 5, 0, 7, 8], idiag=[0, 2, 3, 5, 7], iurow=[1, 3, 4, 6, 8],
                  is_internal=True)),
    dict(name="N5_empty_rows", m=5, n=5,
         row_ptr=[0, 2, 2, 3, 5, 5], col_ind=[3, 0, 2, 1, 4], val=[2, 1, 4, 5, 7],
         exp=dict(icrow=[0, 2, 3, 4, 7, 8], icol=[0, 3, 1, 2, 1, 3, 4, 4],
                  aval=[1, 2, 0, 4, 5, 0, 7, 0], idiag=[0, 2, 3, 5, 7], iurow=[1, 3, 4, 6, 8],
                  is_internal=True)),
    dict(name="N10_random", m=10, n=10,
         row_ptr=[0, 3, 6, 8, 18, 24, 27, 31, 36, 38, 42], col_ind=N10_in_col, val=N10_in_val,
         exp=dict(icrow=[0, 4, 8, 11, 21, 28, 31, 36, 42, 45, 50], icol=N10_out_col,
                  aval=N10_out_val, idiag=[0, 4, 9, 14, 24, 29, 34, 39, 44, 49],
                  iurow=[1, 5, 10, 15, 25, 30, 35, 40, 45, 50], is_internal=True)),
    dict(name="M5_rect_N7", m=5, n=7,
         row_ptr=[0, 3, 5, 6, 10, 13], col_ind=[0, 3, 5, 1, 5, 2, 1, 3, 4, 6, 4, 5, 6],
         val=[1, 2, 1, 3, 2, 4, 5, 6, 7, 3, 8, 4, 5],
         exp=dict(icrow=[0, 3, 5, 6, 10, 13], icol=[0, 3, 5, 1, 5, 2, 1, 3, 4, 6, 4, 5, 6],
                  aval=[1, 2, 1, 3, 2, 4, 5, 6, 7, 3, 8, 4, 5], idiag=[0, 3, 5, 7, 10],
                  iurow=[1, 4, 6, 8, 11], is_internal=False)),
    dict(name="M5_rect_N7_2holes", m=5, n=7,
         row_ptr=[0, 3, 5, 5, 9, 11], col_ind=[0, 3, 5, 1, 5, 1, 3, 4, 6, 5, 6],
         val=[1, 2, 1, 3, 2, 5, 6, 7, 3, 4, 5],
         exp=dict(icrow=[0, 3, 5, 6, 10, 13], icol=[0, 3, 5, 1, 5, 2, 1, 3, 4, 6, 4, 5, 6],
                  aval=[1, 2, 1, 3, 2, 0, 5, 6, 7, 3, 0, 4, 5], idiag=[0, 3, 5, 7, 10],
                  iurow=[1, 4, 6, 8, 11], is_internal=True)),
    dict(name="M7_rect_N5", m=7, n=5,
         row_ptr=[0, 2, 3, 4, 7, 8, 10, 12], col_ind=[0, 3, 1, 2, 1, 3, 4, 4, 1, 2, 0, 3],
         val=[1, 2, 3, 4, 5, 6, 7, 8, 1, 2, 3, 4],
         exp=dict(icrow=[0, 2, 3, 4, 7, 8, 10, 12], icol=[0, 3, 1, 2, 1, 3, 4, 4, 1, 2, 0, 3],
                  aval=[1, 2, 3, 4, 5, 6, 7, 8, 1, 2, 3, 4], idiag=[0, 2, 3, 5, 7],
                  iurow=[1, 3, 4, 6, 8], is_internal=False)),
    dict(name="M7_rect_N5_2holes", m=7, n=5,
         row_ptr=[0, 2, 3, 3, 6, 6, 8, 10], col_ind=[0, 3, 1, 3, 1, 4, 1, 2, 0, 3],
         val=[1, 2, 3, 6, 5, 7, 1, 2, 3, 4],
         exp=dict(icrow=[0, 2, 3, 4, 7, 8, 10, 12], icol=[0, 3, 1, 2, 1, 3, 4, 4, 1, 2, 0, 3],
                  aval=[1, 2, 3, 0, 5, 6, 7, 0, 1, 2, 3, 4], idiag=[0, 2, 3, 5, 7],
                  iurow=[1, 3, 4, 6, 8], is_internal=True)),
]
for c in out["clean_csr"]:
    c["src"] = "tests/unit_tests/common_data_utils.h:609-760 + hint_tests.cpp:75-170"

# ------------------------------------------------------------------------------------------
# TRSV known answers (real double).  variants: (label, fill, trans, unit)
# D7: common_data_utils.h:1373-1543;  S7: :1545-1801;  N25: :1803-2305
# xref in the JSON is already multiplied by alpha (the reference does the same, :1541, :1768).
# ------------------------------------------------------------------------------------------
VARIANTS = [("Lx", "lower", "n", False), ("LL_Ix", "lower", "n", True),
            ("LTx", "lower", "t", False), ("LL_ITx", "lower", "t", True),
            ("Ux", "upper", "n", False), ("UU_Ix", "upper", "n", True),
            ("UTx", "upper", "t", False), ("UU_ITx", "upper", "t", True)]

trsv = []
d7 = dict(m=7, row_ptr=[0, 1, 2, 3, 4, 5, 6, 7], col_ind=[0, 1, 2, 3, 4, 5, 6],
          val=[-2, -4, 3, 5, -7, 9, 4], b=[1, -2, 8, 5, -1, 11, 3], alpha=-9.845233)
for lab, fill, tr, unit in VARIANTS:
    xs = [1, -2, 8, 5, -1, 11, 3] if unit else [-0.5, 0.5, 8.0 / 3.0, 1.0, 1.0 / 7.0, 11.0 / 9.0, 0.75]
    trsv.append(dict(name="D7_" + lab + "_aB", src="tests/unit_tests/common_data_utils.h:1373-1543",
                     fill=fill, trans=tr, unit=unit, xref=[d7["alpha"] * v for v in xs], **d7))

s7 = dict(m=7, row_ptr=[0, 5, 10, 15, 21, 26, 30, 34],
          col_ind=[0, 1, 4, 5, 6, 0, 1, 2, 3, 5, 1, 2, 3, 4, 6, 0, 2,
                   3, 4, 5, 6, 1, 2, 3, 4, 5, 0, 2, 3, 5, 2, 3, 4, 6],
          val=[-2, 1, 3, 7, -1, 2, -4, 1, 2, 4, 6, -2, 9, 1, 9, -9, 1, -2, 1, 1, 1,
               8, 2, 1, -2, 2, 8, 4, 3, 7, 3, 6, 9, 2],
          b=[1, -2, 0, 2, -1, 0, 3], alpha=1.3334)
s7x = {"Lx": [-0.5, 0.25, 0.75, 1.625, 3.0625, -0.553571428571, -18.28125],
       "LL_Ix": [1, -4, 24, -13, -4, -65, 45],
       "LTx": [2.03125, 34.59375, 13.0625, 7.125, 7.25, 0, 1.5],
       "LL_ITx": [85, 12, 35, 12, -28, 0, 3],
       "Ux": [0.625, 2.25, 7.0, 0.0, 0.5, 0.0, 1.5],
       "UU_Ix": [-17, 24, -26, 0, -1, 0, 3],
       "UTx": [-5.0e-1, 3.75e-1, 1.875e-1, 2.1875e-1, -4.6875e-2, 2.6785714e-1, 2.96875e-1],
       "UU_ITx": [1, -3, 3, -19, 12, 0, -4]}
for lab, fill, tr, unit in VARIANTS:
    trsv.append(dict(name="S7_" + lab + "_aB", src="tests/unit_tests/common_data_utils.h:1545-1801",
                     fill=fill, trans=tr, unit=unit, xref=[s7["alpha"] * v for v in s7x[lab]], **s7))


def _nums(txt):
    txt = re.sub(r"\(T\)", "", txt)
    return [float(t) for t in re.findall(r"[-+]?\d+\.?\d*(?:[eE][-+]?\d+)?", txt)]


def extract_n25():
    """Numeric extraction of the N25 system from the reference's fixture database."""
    src = open(os.path.join(REF, "common_data_utils.h")).read()
    lo = src.index("case N25_Lx_aB: // large m test set")
    hi = src.index("case A_nullptr:", lo)
    blk = src[lo:hi]
    res = {}
    for lab, _, _, _ in VARIANTS:
        key = "case N25_%s_aB:" % lab
        # second occurrence = the inner switch (first is the outer case list)
        p = [m.start() for m in re.finditer(re.escape(key), blk)][-1]
        q = blk.index("if constexpr(!cplx)", p)
        a = blk.index("{", q)
        z = blk.index("};", a)
        b = _nums(blk[a + 1:z])
        assert len(b) == 25, (lab, len(b))
        res[lab] = b
    p2 = blk.index("START PART 2")

    def arr(name):
        a = blk.index(name + " = {", p2) if (name + " = {") in blk[p2:] else blk.index(name + "  = {", p2)
        a = blk.index("{", a)
        z = blk.index("};", a)
        return _nums(blk[a + 1:z])

    icrow = [int(v) for v in arr("icrowa")]
    icol = [int(v) for v in arr("icola")]
    aval = arr("aval")
    assert len(icrow) == 26 and len(icol) == 565 and len(aval) == 565, (len(icrow), len(icol), len(aval))
    return icrow, icol, aval, res


n25_path = os.path.join(HERE, "reference_kats.json")
if os.path.isdir(REF):
    icrow, icol, aval, bmap = extract_n25()
    for lab, fill, tr, unit in VARIANTS:
        trsv.append(dict(name="N25_" + lab + "_aB", src="tests/unit_tests/common_data_utils.h:1803-2305",
                         fill=fill, trans=tr, unit=unit, m=25, row_ptr=icrow, col_ind=icol, val=aval,
                         b=bmap[lab], alpha=2.0, xref=[3.0] * 25))
else:  # keep what the committed file already has
    old = json.load(open(n25_path))
    trsv += [t for t in old["trsv"] if t["name"].startswith("N25_")]
out["trsv"] = trsv
# absolute tolerance the reference applies to these systems: expected_precision(10) =
# 10*sqrt(2*eps) (library/src/extra/aoclsparse_utils.hpp:556-580, trsv_tests.cpp:174-186)
out["trsv_abs_tol"] = 10.0 * (2.0 * 2.220446049250313e-16) ** 0.5

# ------------------------------------------------------------------------------------------
# csrmm known answers, tests/unit_tests/csrmm_tests.cpp:99-325 (ids 0, 1, 3; op = none; base 0)
# dense operands are packed (ld = leading dimension of the stated order).
# ------------------------------------------------------------------------------------------
B1 = [1.0, -2.0, 3.0, 4.0, 5.0, -6.0, 1.0, -2.0, 3.0, 4.0, 5.0, -6.0, 1.0,
      -2.0, 3.0, 4.0, 5.0, -6.0, 1.0, -2.0, 3.0, 4.0, 5.0, -6.0, 10]
out["csrmm"] = [
    dict(name="id0_alpha0", src="tests/unit_tests/csrmm_tests.cpp:99-163", m=3, k=3, n=3,
         row_ptr=[0, 2, 3, 4], col_ind=[1, 2, 0, 2], val=[42.0, 0.2, 4.6, -8], alpha=0.0, beta=-3.2,
         B=[-1.0, -2.7, 3.0, 4.5, 5.8, -6.0, 1.0, -2.0, 3.0],
         C=[1.0, -2.0, 3.0, 4.0, 5.0, -6.0, 1.0, -2.0, 3.0],
         C_exp_col=[-3.2, 6.4, -9.6, -12.8, -16, 19.2, -3.2, 6.4, -9.6],
         C_exp_row=[-3.2, 6.4, -9.6, -12.8, -16, 19.2, -3.2, 6.4, -9.6]),
    dict(name="id1_5x5", src="tests/unit_tests/csrmm_tests.cpp:164-217", m=5, k=5, n=5,
         row_ptr=[0, 2, 3, 4, 5, 8], col_ind=[1, 3, 1, 4, 2, 2, 3, 4],
         val=[42.0, 2, 4, 8, 10, 12, 14, 16], alpha=3.0, beta=2.5, B=B1, C=B1,
         C_exp_col=[-225.5, -29, 127.5, 100, 528.5, 129, 14.5, 91, -52.5, 256,
                    -755.5, -87, 74.5, 25, 103.5, 646, 72.5, -63, -177.5, -275,
                    475.5, 58, 252.5, 135, 433],
         C_exp_row=[-729.5, 151, -280.5, 394, 504.5, -87, 14.5, -29, 43.5, 58,
                    84.5, 81, 122.5, -149, 247.5, 160, -167.5, 15, -57.5, 85,
                    499.5, 196, 36.5, -333, 529],
         C_exp_col_T=[2.5, 97, 307.5, 226, 324.5, -15, -741.5, 229, 139.5, 154,
                      12.5, 543, 50.5, 151, 175.5, 10, 576.5, -57, -57.5, -245,
                      7.5, 436, 192.5, 423, 625],
         C_exp_row_T=[2.5, -5, 7.5, 10, 12.5, 39, -237.5, 349, 547.5, 688,
                      240.5, 279, 2.5, -191, 307.5, 142, 168.5, 213, -225.5, 445,
                      271.5, 58, 276.5, -351, 577]),
    dict(name="id3_4x3", src="tests/unit_tests/csrmm_tests.cpp:292-324", m=4, k=3, n=2,
         row_ptr=[0, 0, 1, 2, 3], col_ind=[0, 1, 2], val=[2, 4, 8], alpha=-4.5, beta=11.0,
         B=[3.0, 7.0, 3.0, 1.0, 5.0, 2.0], C=[0.0] * 8,
         C_exp_col=[0, -27, -126, -108, 0, -9, -90, -72],
         C_exp_row=[0, 0, -27, -63, -54, -18, -180, -72]),
]

# ------------------------------------------------------------------------------------------
# symmetric Gauss-Seidel: systems tests/unit_tests/common_data_utils.h:2673-3895, run by
# tests/unit_tests/symgs_tests.cpp:380-455 (x0 = 1, one sweep unless iters says otherwise; the same
# x_gold for both fill modes and both operations of the symmetric systems).  y_gold = A * x_gold as
# held for the ?symgs_mv twin of each system.
# ------------------------------------------------------------------------------------------
def extract_symgs():
    src = open(os.path.join(REF, "common_data_utils.h")).read()

    def block(name, nxt):
        lo = src.index("    case GS_%s:\n    case GS_MV_%s:" % (name, name))
        return src[lo:src.index("    case %s:" % nxt, lo + 10)]

    def real_arrays(blk, name):
        """every `name = {...};` initialiser of the real-typed branches (no bcd( complex literals)"""
        res = []
        for m_ in re.finditer(r"\b%s\s*=\s*\{" % name, blk):
            z = blk.index("};", m_.end())
            body = blk[m_.end():z]
            if "bcd(" not in body:
                res.append(_nums(body))
        return res

    cases = []
    for name, nxt, alpha, iters, lines in (("S7", "GS_TRIDIAG_M5", 1.0, 1, "2673-2813"),
                                           ("TRIDIAG_M5", "GS_BLOCK_TRDIAG_S9", 1.0, 1, "2815-2932"),
                                           ("BLOCK_TRDIAG_S9", "GS_CONVERGE_S4", 1.0, 1, "2934-3084"),
                                           ("CONVERGE_S4", "GS_NONSYM_S4", 1.0, 8, "3086-3202"),
                                           ("SYMM_ALPHA2_S9", "EXT_G5", 2.0, 1, "3743-3895")):
        blk = block(name, nxt)
        rp = [int(v) for v in real_arrays(blk, "icrowa")[0]]
        ci = [int(v) for v in real_arrays(blk, "icola")[0]]
        av = real_arrays(blk, "aval")[0]
        x0 = real_arrays(blk, "x")[0]
        b = real_arrays(blk, "b")[0]
        xg = real_arrays(blk, "wcolxref")[0]
        yg = real_arrays(blk, "xref")[0]
        n = len(rp) - 1
        assert len(ci) == rp[-1] == len(av) and len(b) == len(xg) == len(yg) == len(x0) == n, (name, n)
        cases.append(dict(name="GS_" + name, src="tests/unit_tests/common_data_utils.h:" + lines, mtype="symmetric",
                          m=n, row_ptr=rp, col_ind=ci, val=av, alpha=alpha, iters=iters, x0=x0, b=b,
                          x_gold=xg, y_gold=yg))
    return cases


if os.path.isdir(REF):
    symgs = extract_symgs()
else:
    symgs = [c for c in json.load(open(n25_path))["symgs"] if c["mtype"] == "symmetric"]
# non-symmetric 4x4, general descriptor (common_data_utils.h:3204-3351): small integers, re-typed
symgs.append(dict(name="GS_NONSYM_S4", src="tests/unit_tests/common_data_utils.h:3204-3351", mtype="general", m=4,
                  row_ptr=[0, 4, 8, 12, 16], col_ind=[0, 1, 2, 3] * 4,
                  val=[2, 1, 2, 1, 6, -6, 6, 12, 4, 3, 3, -3, 2, 2, -1, 1], alpha=1.0, iters=1, x0=[1.0] * 4,
                  b=[6, 36, -1, 10],
                  x_gold=dict(n=[-35, 35.333333333333336, 13.666666666666666, 13.333333333333332],
                              t=[-406.16666666666669, 60.5, 53.333333333333336, 121]),
                  y_gold=dict(n=[6, -180, -33, 0.3333333333333286],
                              t=[6, -367.16666666666669, -410.33333333333337, 280.83333333333326])))
out["symgs"] = symgs

# iterative solvers: tests/examples/sample_itsol_d_cg.cpp:62-98 (8x8 SPD, lower triangle stored, x0 = 1, "CG Abs
# Tolerance" 5e-6, "CG Preconditioner" SGS, b = A * expected) and tests/examples/sample_itsol_d_gmres.cpp:100-116
# (cage4, 9x9 unsymmetric)
out["itsol"] = dict(
    cg=dict(src="tests/examples/sample_itsol_d_cg.cpp:62-98", n=8, row_ptr=[0, 1, 2, 5, 6, 8, 11, 15, 18],
            col_ind=[0, 1, 0, 1, 2, 3, 1, 4, 0, 4, 5, 0, 3, 4, 6, 2, 5, 7],
            val=[19, 10, 1, 8, 11, 13, 2, 11, 2, 1, 9, 7, 9, 5, 12, 5, 5, 9], x0=[1.0] * 8,
            expected=[1.0, 0.0, 1.0, 0.0, 1.0, 0.0, 1.0, 0.0], abs_tol=5.0e-6, precond="SGS"),
    gmres=dict(src="tests/examples/sample_itsol_d_gmres.cpp:100-116", n=9,
               row_ptr=[0, 5, 10, 15, 20, 26, 32, 38, 44, 49],
               col_ind=[0, 1, 3, 4, 7, 0, 1, 2, 4, 5, 1, 2, 3, 5, 6, 0, 2, 3, 6, 7, 0, 1, 4, 5, 6,
                        8, 1, 2, 4, 5, 7, 8, 2, 3, 4, 6, 7, 8, 0, 3, 5, 6, 7, 8, 4, 5, 6, 7, 8],
               val=[0.75, 0.14, 0.11, 0.14, 0.11, 0.08, 0.69, 0.11, 0.08, 0.11, 0.09, 0.67, 0.08,
                    0.09, 0.08, 0.09, 0.14, 0.73, 0.14, 0.09, 0.04, 0.04, 0.54, 0.14, 0.11, 0.25,
                    0.05, 0.05, 0.08, 0.45, 0.08, 0.15, 0.04, 0.04, 0.09, 0.47, 0.09, 0.18, 0.05,
                    0.05, 0.14, 0.11, 0.55, 0.25, 0.08, 0.08, 0.09, 0.08, 0.17]))

# ELL: tests/unit_tests/ellmv_tests.cpp:151-252 (3x3, one-based CSR converted with csr2ell, and the same
# matrix given directly as one-based ELL with -1 padding)
out["ell"] = [dict(name="M3_base1", src="tests/unit_tests/ellmv_tests.cpp:151-252", base=1, m=3, n=3,
                   row_ptr=[1, 2, 3, 5], col_ind=[1, 2, 1, 3], val=[8.0, 5.0, 7.0, 7.0], x=[1.0, 2.0, 3.0],
                   alpha=1.0, beta=0.0, ell_width=2, ell_col_ind=[1, -1, 2, -1, 1, 3],
                   ell_val=[8.0, 0.0, 5.0, 0.0, 7.0, 7.0], y_gold=[8.0, 10.0, 28.0])]

# BLKCSR: tests/unit_tests/blkcsrmv_tests.cpp:444-470 (block arrays given directly, one-based, 2x8 blocks),
# :518-537 (one-based CSR through csr2blkcsr, rows_blk 1/2/4) and :656-676 (the same, zero-based)
_blk_common = dict(m=6, n=8, nnz=14, x=[1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 7.0, 8.0], alpha=1.0, beta=0.0,
                   val=[8.0, 2.0, 3.0, 3.0, 3.0, 6.0, 10.0, 9.0, 6.0, 2.0, 2.0, 3.0, 2.0, 6.0],
                   y_gold=[8.0, 2.0, 0.0, 204.0, 23.0, 14.0])
out["blkcsr"] = dict(
    direct=dict(_blk_common, src="tests/unit_tests/blkcsrmv_tests.cpp:444-470", base=1, rows_blk=2,
                blk_col_ind=[1, 1, 1], blk_row_ptr=[1, 2, 2, 3, 3, 4, 4], masks=[1, 1, 0, 255, 24, 3]),
    csr=[dict(_blk_common, src="tests/unit_tests/blkcsrmv_tests.cpp:518-537", base=1,
              row_ptr=[1, 2, 3, 3, 11, 13, 15], col_ind=[1, 1, 1, 2, 3, 4, 5, 6, 7, 8, 4, 5, 1, 2]),
         dict(_blk_common, src="tests/unit_tests/blkcsrmv_tests.cpp:656-676", base=0,
              row_ptr=[0, 1, 2, 2, 10, 12, 14], col_ind=[0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 3, 4, 0, 1])])

# ------------------------------------------------------------------------------------------
# Dense-result product and CSR -> dense: tests/unit_tests/spmmd_tests.cpp:130-157 (real 3x3 case: C = A*B and
# C = A^T*B, row-major 3x3) and tests/unit_tests/conversion_tests.cpp:211-252 (5x5, 16 non-zeros; row-major,
# column-major, and row-major with ld = 8 whose padding stays 0)
# ------------------------------------------------------------------------------------------
out["spmmd"] = dict(src="tests/unit_tests/spmmd_tests.cpp:130-157", m=3, k=3, n=3,
                    a=dict(row_ptr=[0, 2, 3, 6], col_ind=[0, 2, 2, 0, 1, 2], val=[1, 2, 3, 4, 5, 6]),
                    b=dict(row_ptr=[0, 2, 3, 4], col_ind=[0, 1, 2, 1], val=[1, 2, 3, 4]),
                    c_none=[1, 10, 0, 0, 12, 0, 4, 32, 15], c_trans=[1, 18, 0, 0, 20, 0, 2, 28, 9])
out["csr2dense"] = dict(src="tests/unit_tests/conversion_tests.cpp:211-252", m=5, n=5,
                        row_ptr=[0, 3, 6, 10, 12, 16], col_ind=[0, 1, 4, 1, 2, 4, 0, 1, 2, 3, 2, 3, 0, 1, 2, 4],
                        val=[1, 1, 4, 2, 4, 1, 2, 1, 8, 2, 4, 1, 3, 6, 2, 1],
                        rowmajor=[1, 1, 0, 0, 4, 0, 2, 4, 0, 1, 2, 1, 8, 2, 0, 0, 0, 4, 1, 0, 3, 6, 2, 0, 1],
                        colmajor=[1, 0, 2, 0, 3, 1, 2, 1, 0, 6, 0, 4, 8, 4, 2, 0, 0, 2, 1, 0, 4, 1, 0, 0, 1],
                        rowmajor_ld8=[1, 1, 0, 0, 4, 0, 0, 0, 0, 2, 4, 0, 1, 0, 0, 0, 2, 1, 8, 2,
                                      0, 0, 0, 0, 0, 0, 4, 1, 0, 0, 0, 0, 3, 6, 2, 0, 1, 0, 0, 0])

# ------------------------------------------------------------------------------------------
# Level 1 (real types): tests/unit_tests/axpyi_tests.cpp:78-88, roti_tests.cpp:50-99, dotp_tests.cpp:44,64-69,
# gthr_tests.cpp:42-43,66-71, sctr_tests.cpp:71-84
# ------------------------------------------------------------------------------------------
out["level1"] = dict(
    axpyi=dict(src="tests/unit_tests/axpyi_tests.cpp:78-88", a=30, x=[1, 2, 3, 4], indx=[3, 6, 8, 0],
               y=[31, 0, 0, 0, 0, 0, 0, 0, 7], y_nnz4=[151, 0, 0, 30, 0, 0, 60, 0, 97], y_nnz2=[31, 0, 0, 30, 0, 0, 60, 0, 7]),
    roti=[dict(src="tests/unit_tests/roti_tests.cpp:50-99", c=c, s=s_, indx=ix, x=x, y=y, x_exp=xe, y_exp=ye) for c, s_, ix, x, y, xe, ye in [
        (-2, 2, [0, 3, 6], [1, 4, 8], [1, 0, 0, 4, 0, 0, 8], [0, 0, 0], [-4, 0, 0, -16, 0, 0, -32]),
        (-4.5, 3.5, [0, 3, 6], [1, 4, 8], [1, 0, 0, 4, 0, 0, 8], [-1, -4, -8], [-8, 0, 0, -32, 0, 0, -64]),
        (-4.5, 3.5, [0, 3, 6], [4.75, -2.5, 7], [4.75, 0, 0, -2.5, 0, 0, 7], [-4.75, 2.5, -7], [-38, 0, 0, 20, 0, 0, -56]),
        (-4.5, 3.5, [0, 3, 6, 7, 9], [-0.75, 4, -9.5, 46, 1.25], [-0.75, 0, 0, 4, 0, 0, -9.5, 46, 0, 1.25],
         [0.75, -4, 9.5, -46, -1.25], [6, 0, 0, -32, 0, 0, 76, -368, 0, -10]),
        (2, 2, [0, 3, 6, 7, 9], [-0.75, 4, -9.5, 46, 1.25], [-0.75, 0, 0, 4, 0, 0, -9.5, 46, 0, 1.25],
         [-3, 16, -38, 184, 5], [0, 0, 0, 0, 0, 0, 0, 0, 0, 0])]],
    doti=dict(src="tests/unit_tests/dotp_tests.cpp:44,64-69", indx=[6, 1, 4, 20, 2, 3, 7, 8, 10, 12, 13, 15, 16, 18, 0, 14, 5, 11],
              x=[1, 0, 3, 4, 0, 0, 7, 8, 0, 10, 0, 12, 0, 0, 15, 16, 0, 18],
              y=[-4.7, 2, -1.3, 5, 4, 3, 1, 6, -7, 12, -3, 0.5, 4.5, 3.5, 15, 2, 8, 2, 9, 10, 6.25], dot=271.5),
    gthr=dict(src="tests/unit_tests/gthr_tests.cpp:42-43,66-71", indx=[0, 3, 5, 1, 7, 12, 2, 6, 8, 9, 10, 11, 4, 13, 15, 16, 14, 18],
              y=list(range(1, 23)), x_exp=[1, 4, 6, 2, 8, 13, 3, 7, 9, 10, 11, 12, 5, 14, 16, 17, 15, 19],
              y_gthrz=[0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 18, 0, 20, 21, 22]),
    sctr=dict(src="tests/unit_tests/sctr_tests.cpp:71-84", indx=[1, 5, 13, 14, 6, 8, 9, 3, 7, 2, 10, 0, 15, 12, 4, 11, 16],
              x=list(range(1, 18)), y_nnz17=[12, 1, 10, 8, 15, 2, 5, 9, 6, 7, 11, 16, 14, 3, 4, 13, 17],
              y_nnz10=[0, 1, 10, 8, 0, 2, 5, 9, 6, 7, 0, 0, 0, 3, 4, 0, 0]))

# ------------------------------------------------------------------------------------------
# DIA / BSR: tests/unit_tests/diamv_tests.cpp:137-197 (one-based CSR -> DIA -> diamv) and
# tests/unit_tests/bsrmv_tests.cpp:40-101 (zero-based CSR -> BSR, block 2, column-major blocks -> bsrmv; y padded to 6)
# ------------------------------------------------------------------------------------------
out["dia_bsr"] = dict(src="tests/unit_tests/diamv_tests.cpp:137-197, bsrmv_tests.cpp:40-101", m=5, n=5,
                      row_ptr=[0, 1, 2, 4, 6, 7], col_ind=[0, 1, 1, 2, 0, 3, 3], val=[6, 1, 2, 3, 5, 1, 10],
                      x=[1, 2, 3, 4, 5, 6], y_gold=[6, 2, 13, 9, 40, 0], bsr_dim=2)

# ------------------------------------------------------------------------------------------
# Forward SOR sweep: tests/unit_tests/sorv_tests.cpp:366-414 (octave-generated x after 1 and 10 sweeps, and with
# alpha = 0 applied to the 10-sweep iterate) and tests/examples/sample_dsorv.cpp:51-61
# ------------------------------------------------------------------------------------------
out["sorv"] = [
    dict(src="tests/unit_tests/sorv_tests.cpp:366-414", n=4, row_ptr=[0, 3, 7, 10, 13], col_ind=[0, 1, 2, 0, 1, 2, 3, 1, 2, 3, 0, 2, 3],
         val=[4.0, -1.0, -6.0, -5.0, -4.0, 10.0, 8.0, 9.0, 4.0, -2.0, 1.0, -7.0, 5.0], b=[2.0, 21.0, -12.0, -6.0],
         x0=[1.0, -0.5, -2.0, 3.7], omega=0.5,
         x_iter1=[-0.8125, -1.1671874999999998, -0.26191406249999982, 1.14791015625],
         x_iter10=[2.8668745958572917, -2.0001324279196497, 1.9725100983350874, 0.97782651833978285],
         x_iter10_then_alpha0=[0.25, -2.78125, 1.62890625, 0.51523437500000002]),
    dict(src="tests/examples/sample_dsorv.cpp:51-61", n=4, row_ptr=[0, 2, 5, 6, 10], col_ind=[0, 1, 0, 1, 2, 2, 1, 3, 2, 0],
         val=[111.1, 2.345, 3.12, 9.87, -56.2, -39.678, 76.9, -25.106, -903.40, 32.0987],
         b=[157.5045, -489.033, -321.3918, -7433.72955], x0=[1, -41, 5.7, 0.341], omega=0.5,
         x_iter1=[1.6415369036903691, -29.305197322163828, 6.9000000000000004, -19.757185084519875]),
]

# ------------------------------------------------------------------------------------------
# aoclsparse_?mv with a TRIANGULAR descriptor on a rectangular 5 x 4 matrix: tests/unit_tests/mv_tests.cpp:344-388
# (test_mv_success: exp_y_l with fill = lower, exp_y_u with fill = upper; alpha = 1, beta = 0, op = none, base 0)
# ------------------------------------------------------------------------------------------
out["mv_tri"] = dict(src="tests/unit_tests/mv_tests.cpp:344-388", base=0, m=5, n=4, row_ptr=[0, 2, 3, 4, 7, 8],
                     col_ind=[0, 3, 1, 2, 1, 2, 3, 1], val=[1, 2, 3, 4, 5, 6, 7, 8], x=[1.0, 2.0, 3.0, 4.0],
                     alpha=1.0, beta=0.0, exp_y_l=[1, 6, 12, 56, 16], exp_y_u=[9, 6, 12, 28, 0])

with open(n25_path, "w") as f:
    json.dump(out, f, indent=None, separators=(",", ":"))
    f.write("\n")
print("wrote", n25_path, os.path.getsize(n25_path), "bytes;",
      len(out["trsv"]), "trsv systems,", len(out["symgs"]), "symgs systems")
