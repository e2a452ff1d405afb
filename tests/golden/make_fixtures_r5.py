#!/usr/bin/env python3
"""Generate tests/golden/reference_kats_r5.json: the extreme-value known answers of the reference's
tests/unit_tests/extreme_value_tests.cpp (NaN / Inf / max / min propagation through aoclsparse_add, aoclsparse_sp2m,
aoclsparse_?csrmm and the sparse dot product) -- the one reference test file for the hot path (sp2m, csrmm) that rounds 1-4 had
not transcribed.  Data only (inputs + expected outputs), extracted numerically from the file where it lies (this container
only); nothing under tests/ reads /root/reference at run time.

  :33-87     init(): A (7 x 7, 17 entries: NaN, Inf, max, min among ones), B as CSR (21 entries) and the same B dense, row-major
  :89-180    add:   C = A + B, expected values (compared in sorted-CSR order, values only)
  :182-305   sp2m:  C = A * B (full computation), expected values
  :307-432   csrmm: C = 1 * A * B_dense + 0 * C, row-major, kid 1 (AVX2) or 3 (AVX-512), expected 7 x 7
  :434-482   dot:   five configurations of x (NaN; Inf - Inf; Inf; overflow; max) with their expected results
Special values are written as strings: "nan", "inf", "-inf", "max", "min" (the type's largest / smallest normal) and "tmp" =
max * 8.9885e-24 + 1 evaluated in the type (:198, :325); the check the reference applies is EXPECT_ARR_MATCH with one ulp-scale
tolerance (NaN matches NaN, Inf matches Inf of the same sign).

Run:  python tests/golden/make_fixtures_r5.py
"""
import json
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/tests/unit_tests/extreme_value_tests.cpp"


def tokens(body):
    """initializer list -> numbers and the special-value names"""
    out = []
    for raw in body.replace("\n", " ").split(","):
        t = re.sub(r"/\*.*?\*/", "", raw).strip()
        if not t:
            continue
        neg = t.startswith("-")
        u = t.lstrip("-")
        if "quiet_NaN" in u:
            out.append("nan")
        elif "infinity" in u:
            out.append("-inf" if neg else "inf")
        elif "::max)()" in u:
            out.append("max")
        elif "::min)()" in u:
            out.append("min")
        elif u == "tmp":
            out.append("tmp")
        else:
            out.append(float(t))
    return out


def assigns(text, name):
    return [tokens(m.group(1)) for m in re.finditer(r"\b%s\.assign\(\s*\{(.*?)\}\)" % re.escape(name), text, re.S)]


def extract_symm_opt():
    """tests/unit_tests/optimize_symm_herm_tests.cpp:39-758 (generate_test_matrix, real types): four small matrices (one with
    unsorted rows and missing diagonal entries) and, per triangle, the symmetric matrix the reference's optimize builds from it --
    both triangles, a diagonal entry in every row -- with the values it holds for non-unit / unit / zero diagonal types.  The
    reference compares these with its internal CSR copy after set_mv_hint + optimize (+ one mv) for every (fill, base, op, diag)
    (:760-938); through the public interface they say what y = op(A) x must be."""
    path = "/root/reference/tests/unit_tests/optimize_symm_herm_tests.cpp"
    src = open(path).read().split("\n")
    tops = [i for i, l in enumerate(src) if re.match(r"        case \d+:$", l)]
    tops.append(next(i for i, l in enumerate(src) if "void test_opt_symm_herm_matrix" in l))
    nums = lambda t: [float(x) for x in re.findall(r"[-+]?\d+\.?\d*(?:[eE][-+]?\d+)?", t)]
    out = []
    for k in range(len(tops) - 1):
        blk = src[tops[k]:tops[k + 1]]
        head = "\n".join(blk[:next(i for i, l in enumerate(blk) if "switch(doid)" in l)])
        mm = re.search(r"m = (\d+), n = (\d+), nnz = (\d+);", head)
        ent = {"id": k, "src": "tests/unit_tests/optimize_symm_herm_tests.cpp:%d-%d" % (tops[k] + 1, tops[k + 1]),
               "m": int(mm.group(1)), "n": int(mm.group(2)), "nnz": int(mm.group(3)),
               "col_ind": [int(x) for x in nums(re.search(r"col_ind\.assign\(\{(.*?)\}\)", head, re.S).group(1))],
               "row_ptr": [int(x) for x in nums(re.search(r"row_ptr\.assign\(\{(.*?)\}\)", head, re.S).group(1))],
               "val": nums(re.search(r"\bval\.assign\(\{(.*?)\}\)", head, re.S).group(1)), "expected": {}}
        idx = [i for i, l in enumerate(blk) if re.match(r"\s*case aoclsparse::doid::\w+:", l)]
        groups, cur = [], None
        for i in idx:
            name = re.search(r"doid::(\w+)", blk[i]).group(1)
            if cur and i == cur["last"] + 1:
                cur["names"].append(name), cur.update(last=i)
            else:
                cur = {"names": [name], "first": i, "last": i}
                groups.append(cur)
        for gi, g in enumerate(groups):
            t = "\n".join(blk[g["last"] + 1:groups[gi + 1]["first"] if gi + 1 < len(groups) else len(blk)])
            e = {}
            for f in ("ptr", "ind", "idiag", "non_unit_diag_val", "unit_diag_val", "zero_diag_val", "diag_val"):
                body = re.search(r"sol_opt_csr_t\.%s\.assign\(\s*\{(.*?)\}\);" % f, t, re.S).group(1)
                if "{" in body:  # (a hermitian group lists the complex values first: the real branch is not needed here)
                    e = None
                    break
                e[f] = nums(body) if f.endswith("val") else [int(x) for x in nums(body)]
            if e is None:
                continue
            assert len(e["ind"]) == e["ptr"][-1] == len(e["non_unit_diag_val"]) == len(e["unit_diag_val"]) == len(e["zero_diag_val"])
            assert len(e["idiag"]) == ent["m"] == len(e["diag_val"])
            for nm in g["names"]:
                if nm in ("sl", "su"):  # real symmetric: lower / upper triangle of the input
                    ent["expected"]["lower" if nm == "sl" else "upper"] = e
        assert set(ent["expected"]) == {"lower", "upper"}, (k, list(ent["expected"]))
        out.append(ent)
    return {"src": "tests/unit_tests/optimize_symm_herm_tests.cpp:39-938", "zero_based": True, "matrices": out}


def main():
    lines = open(SRC).read().split("\n")

    def span(first_pat, last_pat):
        lo = next(i for i, l in enumerate(lines) if first_pat in l)
        hi = next(i for i, l in enumerate(lines) if i > lo and last_pat in l)
        return lo, hi, "\n".join(lines[lo:hi])

    lo, hi, t_init = span("void init(", "void test_ev_add()")
    kats = {"_about": "reference extreme-value vectors added in round 5 (tests/golden/make_fixtures_r5.py); data only",
            "tmp": "max * 8.9885e-24 + 1 in the type (extreme_value_tests.cpp:198,325)"}
    ints = lambda v: [int(x) for x in v]
    kats["init"] = {"src": "tests/unit_tests/extreme_value_tests.cpp:%d-%d" % (lo + 1, hi), "m": 7, "A_nnz": 17, "B_nnz": 21,
                    "A_val": assigns(t_init, "A_val")[0], "A_col_ind": ints(assigns(t_init, "A_col_ind")[0]),
                    "A_row_ptr": ints(assigns(t_init, "A_row_ptr")[0]), "B_val": assigns(t_init, "B_val")[0],
                    "B_col_ind": ints(assigns(t_init, "B_col_ind")[0]), "B_row_ptr": ints(assigns(t_init, "B_row_ptr")[0]),
                    "B_dense_row_major": assigns(t_init, "B_dense")[0]}
    for name, first, last in (("add", "void test_ev_add()", "void test_ev_sp2m()"), ("sp2m", "void test_ev_sp2m()", "void test_ev_csrmm()"),
                              ("csrmm", "void test_ev_csrmm()", "void test_ev_dot()")):
        lo, hi, t = span(first, last)
        d = {"src": "tests/unit_tests/extreme_value_tests.cpp:%d-%d" % (lo + 1, hi), "C_exp_val": assigns(t, "C_exp_val")[0]}
        if name != "csrmm":
            d["C_exp_col_ind"], d["C_exp_row_ptr"] = ints(assigns(t, "C_exp_col_ind")[0]), ints(assigns(t, "C_exp_row_ptr")[0])
        else:
            d.update(alpha=1.0, beta=0.0, order="row", kid="1 (3 where AVX-512 kernels run)", ldb=7, ldc=7, n=7)
        if name == "sp2m":
            # the product has 34 entries (row 6 = columns 0 2 3 4 5) and 34 expected values; the reference's C_exp_col_ind /
            # C_exp_row_ptr hold 33 (column 4 of the last row is missing) and are never compared (:296-299: values only)
            d["note"] = ("34 expected values = the 34 entries of the product in sorted-CSR order; C_exp_col_ind / C_exp_row_ptr are kept "
                         "as the reference lists them (33 entries, last row one short) and are NOT compared there (values only)")
        kats[name] = d
    lo, hi, t = span("void test_ev_dot()", "TEST(add, EVDouble)")
    x0, y = assigns(t, "x")[0], assigns(t, "y")[0]
    kats["dot"] = {"src": "tests/unit_tests/extreme_value_tests.cpp:%d-%d" % (lo + 1, hi), "nnz": 18, "indx": ints(assigns(t, "indx")[0]),
                   "x": x0, "y": y,
                   # (x[0], x[1]) replaced in turn, expected result (:454-481)
                   "cases": [{"x0": "nan", "x1": 0.0, "expected": "nan"}, {"x0": "inf", "x1": "-inf", "expected": "nan"},
                             {"x0": "inf", "x1": 1.0, "expected": "inf"}, {"x0": "max", "x1": "max", "expected": "inf"},
                             {"x0": "max", "x1": 1.0, "expected": "max"}]}
    kats["symm_opt"] = extract_symm_opt()
    assert len(kats["init"]["A_val"]) == 17 and len(kats["init"]["B_val"]) == 21 and len(kats["init"]["B_dense_row_major"]) == 49
    assert len(kats["add"]["C_exp_val"]) == 26 and len(kats["sp2m"]["C_exp_val"]) == 34 and len(kats["csrmm"]["C_exp_val"]) == 49
    assert len(kats["dot"]["x"]) == 18 and len(kats["dot"]["y"]) == 21
    with open(os.path.join(HERE, "reference_kats_r5.json"), "w") as f:
        json.dump(kats, f, indent=0)
        f.write("\n")
    print("wrote reference_kats_r5.json")


if __name__ == "__main__":
    main()
