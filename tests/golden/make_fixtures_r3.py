#!/usr/bin/env python3
"""Generate tests/golden/reference_kats_r3.json: the reference vectors VERDICT r2 (missing 2 / 6) found absent from
tests/golden/reference_kats.json.  Data only (inputs + expected outputs); extracted numerically from the reference's unit
tests where they lie (this container only) -- nothing under tests/ reads /root/reference at run time.

  csrmm   real-type `init` ids 2, 4, 5, 7, 8, 9, 10, 11 and the padded-ld case of id 6
          tests/unit_tests/csrmm_tests.cpp:218-650 (data), :1808-1830 (ld rule), :2055-2170 (how a case is run:
          set_mm_hint + optimize, C resized to C_m*C_n, EXPECT_DOUBLE_EQ_VEC), :1995-2050 (greater ld), :2174-2660 (which
          (op, order, fill, diag, base, kid) combinations are run for which id)
  csr2m   the base-one 3x3 product with its gold CSR, all three base mixes, one- and two-stage
          tests/unit_tests/csr2m_tests.cpp:216-600
  sp2m    the hand-computed CSC x CSR product (tests/unit_tests/sp2m_tests.cpp:371-444) and the CONFIGURATIONS of the
          randomised success tests (:880-1050: dimensions, nnz, bases, operations, stage, formats; their check is a dense
          product at sqrt(eps), :501-585) -- the reference holds no literal sp2m vectors besides the CSC one
  mv      the extreme-value SpMV cases (tests/unit_tests/mv_tests.cpp:1858-2290) and empty rows after optimize (:1320-1360)

Run:  python tests/golden/make_fixtures_r3.py
"""
import json
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/tests/unit_tests"


def nums(s):
    return [float(t) for t in re.findall(r"[-+]?\d+\.?\d*(?:[eE][-+]?\d+)?", s)]


def extract_csrmm():
    src = open(os.path.join(REF, "csrmm_tests.cpp")).read().split("\n")
    # the real-type switch: from the first "case 0:" to the complex branch
    lo = next(i for i, l in enumerate(src) if l.strip() == "case 0:")
    hi = next(i for i, l in enumerate(src) if i > lo and "std::is_same_v<T, aoclsparse_double_complex>" in l)
    cases, cur = {}, None
    for i in range(lo, hi):
        mm = re.match(r"\s*case (\d+):", src[i])
        if mm:
            cur = int(mm.group(1))
            cases[cur] = dict(first=i + 1, lines=[])
        elif cur is not None:
            cases[cur]["lines"].append(src[i])
            cases[cur]["last"] = i + 1
    out = {}
    for cid, cs in cases.items():
        text = "\n".join(cs["lines"])
        d = dict(id=cid, src="tests/unit_tests/csrmm_tests.cpp:%d-%d" % (cs["first"], cs["last"]))
        for name in ("m", "k", "n", "nnz"):
            d[name] = int(re.search(r"\b%s\s*=\s*(\d+)" % name, text).group(1))
        for name in ("alpha", "beta"):
            d[name] = float(re.search(r"\b%s\s*=\s*([-+]?\d+\.?\d*)" % name, text).group(1))
        for name in ("csr_val", "csr_col_ind", "csr_row_ptr", "B", "C"):
            mm = re.search(r"\b%s\.assign\(\{(.*?)\}\)" % name, text, re.S)
            if mm:
                d[name] = nums(mm.group(1))
            else:
                mm = re.search(r"\b%s\.assign\((\d+), T\{\}\)" % name, text)
                # id 8 never assigns C: run_csrmm_case's fresh vector is resized to zeros (csrmm_tests.cpp:2111)
                d[name] = [0.0] * (int(mm.group(1)) if mm else 1)
        for name in ("csr_col_ind", "csr_row_ptr"):
            d[name] = [int(v) for v in d[name]]
        # expected results keyed by (sym variant, order, op); walk the statements tracking the enclosing conditions
        exp, sym, order, op = {}, None, None, None
        ind_order = ind_op = -1
        lines = cs["lines"]
        j = 0
        while j < len(lines):
            l = lines[j]
            ind = len(l) - len(l.lstrip())
            s = l.strip()
            mm = re.match(r"if\((sym[ul]t_(?:non_)?unit)\)", s)
            if mm:
                sym, order, op = mm.group(1), None, None
            elif s.startswith("if(order == aoclsparse_order_column)"):
                order, op, ind_order = "col", None, ind
            elif s.startswith("else if(order == aoclsparse_order_row)"):
                order, op = "row", None
            elif s.startswith("if(op == aoclsparse_operation_none)"):
                op, ind_op = "n", ind
            elif s.startswith("if(op == aoclsparse_operation_transpose"):
                op = "t"
            elif s == "else":
                if ind == ind_op and order is not None and op is not None:
                    op = "t"
                else:
                    order, op = "row", None
            if "C_exp.assign({" in s:
                buf = s
                while "});" not in buf:
                    j += 1
                    buf += lines[j]
                exp["%s/%s/%s" % (sym or "-", order or "-", op or "-")] = nums(buf[buf.index("{"):])
            j += 1
        d["C_exp"] = exp
        out[cid] = d
    return out


def csrmm_runs(cases):
    """Expand into the (format, type, fill, diag, op, order) combinations the reference runs (csrmm_tests.cpp:2174-2660)."""
    runs = []

    def dims(c, op, order):
        m, k, n = c["m"], c["k"], c["n"]
        A_m, B_m = (m, k) if op == "n" else (k, m)
        ldb, ldc = (B_m, A_m) if order == "col" else (n, n)
        return A_m, B_m, ldb, ldc

    def pick(c, sym, order, op):
        e = c["C_exp"]
        for key in ("%s/%s/%s" % (sym, order, op), "%s/%s/-" % (sym, order), "%s/-/-" % sym):
            if key in e:
                return e[key]
        raise KeyError((c["id"], sym, order, op, sorted(e)))

    def add(c, fmt, mtype, fill, diag, op, order, kid=0):
        A_m, B_m, ldb, ldc = dims(c, op, order)
        sym = "-"
        if mtype == "symmetric":
            sym = "sym%st_%s" % ("u" if fill == "upper" else "l", "non_unit" if diag == "non_unit" else "unit")
        exp = pick(c, sym, order, op)
        nB = B_m * c["n"]
        C0 = (c["C"] + [0.0] * (A_m * c["n"]))[:A_m * c["n"]] if len(c["C"]) > 1 else [0.0] * (A_m * c["n"])
        runs.append(dict(id=c["id"], src=c["src"], format=fmt, type=mtype, fill=fill, diag=diag, op=op, order=order, kid=kid,
                         m=c["m"], k=c["k"], n=c["n"], nnz=c["nnz"], ptr=c["csr_row_ptr"] if fmt == "csr" else c["csr_col_ind"],
                         ind=c["csr_col_ind"] if fmt == "csr" else c["csr_row_ptr"], val=c["csr_val"], alpha=c["alpha"],
                         beta=c["beta"], B=c["B"][:nB], ldb=ldb, C=C0, ldc=ldc, C_exp=exp[:A_m * c["n"]],
                         base=1 if c["id"] == 5 else 0))

    for cid in (2, 4):                                   # general CSR (ids 0, 1, 3 are in reference_kats.json)
        for op in ("n", "t"):
            for order in ("row", "col"):
                add(cases[cid], "csr", "general", "lower", "non_unit", op, order)
    for order in ("row", "col"):                         # id 5: one-based CSR data (used by the invalid-size test as a valid call)
        add(cases[5], "csr", "general", "lower", "non_unit", "n", order)
    for cid in (7, 8):                                   # symmetric CSR
        for fill in ("upper", "lower"):
            for diag in ("non_unit", "unit"):
                for op in ("n", "t"):
                    for order in ("row", "col"):
                        add(cases[cid], "csr", "symmetric", fill, diag, op, order)
    for op in ("n", "t"):                                # general CSC id 9, non-square CSC id 11
        for order in ("row", "col"):
            add(cases[9], "csc", "general", "lower", "non_unit", op, order)
            add(cases[11], "csc", "general", "lower", "non_unit", op, order)
    add(cases[9], "csc", "general", "lower", "non_unit", "n", "row", kid=1)
    for fill in ("upper", "lower"):                      # symmetric CSC id 10
        for diag in ("non_unit", "unit"):
            for op in ("n", "t"):
                for order in ("row", "col"):
                    add(cases[10], "csc", "symmetric", fill, diag, op, order)
    return runs


def main():
    out = {"_about": "reference vectors added in round 3 (see tests/golden/make_fixtures_r3.py); data only"}
    cases = extract_csrmm()
    out["csrmm_runs"] = csrmm_runs(cases)
    c6 = cases[6]
    out["csrmm_greater_ld"] = dict(src="tests/unit_tests/csrmm_tests.cpp:1995-2050", m=c6["m"], k=c6["k"], n=c6["n"],
                                   ptr=c6["csr_row_ptr"], ind=c6["csr_col_ind"], val=c6["csr_val"], alpha=c6["alpha"],
                                   beta=c6["beta"], order="col", ldb=2 * c6["k"], ldc=2 * c6["m"],
                                   B=[1.0, -2.0, 3.0, 0, 0, 0, 4.0, 5.0, -6.0, 0, 0, 0], C=[0.1, 0.2, 0, 0, 0.3, 0.4, 0, 0],
                                   C_exp=[1.12, -190.96, 0, 0, 3.36, 487.48, 0, 0])

    # csr2m, tests/unit_tests/csr2m_tests.cpp:216-600: A (3x3, 4 nnz) * B (3x3, 4 nnz), one-based data, zero-based gold
    out["csr2m"] = dict(src="tests/unit_tests/csr2m_tests.cpp:216-246,397-420", m=3, k=3, n=3,
                        A=dict(ptr=[1, 2, 3, 5], ind=[1, 2, 1, 3], val=[8.0, 5.0, 7.0, 7.0]),
                        B=dict(ptr=[1, 2, 3, 5], ind=[1, 1, 2, 3], val=[7.0, 9.0, 6.0, 2.0]),
                        C=dict(ptr=[0, 1, 2, 5], ind=[0, 0, 0, 1, 2], val=[56.0, 45.0, 49.0, 42.0, 14.0]),
                        base_mixes=[[1, 1], [1, 0], [0, 1]], stages=["full", "two-stage"],
                        invalid_base_status="invalid_value")
    # sp2m: the one literal case + the configurations of the randomised tests
    out["sp2m_csc"] = dict(src="tests/unit_tests/sp2m_tests.cpp:371-444", m=3, n=3,
                           A_csc=dict(ptr=[0, 1, 3, 4], ind=[0, 0, 1, 2], val=[1.0, 3.0, 2.0, 4.0]),
                           B_csr=dict(ptr=[0, 1, 2, 3], ind=[0, 1, 2], val=[1.0, 1.0, 1.0]),
                           dense_C=[1, 3, 0, 0, 2, 0, 0, 0, 4])
    src = open(os.path.join(REF, "sp2m_tests.cpp")).read()
    cfgs = []
    opmap = {"aoclsparse_operation_none": "n", "op_none": "n", "aoclsparse_operation_transpose": "t", "op_trans": "t",
             "aoclsparse_operation_conjugate_transpose": "h", "op_conj": "h"}
    for mm in re.finditer(r"test_sp2m_(success|finalize)<(\w+)>\(([^;]*?)\);", src, re.S):
        kind, typ, args = mm.group(1), mm.group(2), [a.strip() for a in mm.group(3).replace("\n", " ").split(",")]
        if len(args) < 10 or not args[0].isdigit():
            continue
        line = src[:mm.start()].count("\n") + 1
        base = lambda s: 1 if s in ("aoclsparse_index_base_one", "base1") else 0
        cfg = dict(src="tests/unit_tests/sp2m_tests.cpp:%d" % line, kind=kind,
                   type={"double": "d", "float": "s", "aoclsparse_double_complex": "z", "aoclsparse_float_complex": "c"}[typ],
                   m_a=int(args[0]), n_a=int(args[1]), m_b=int(args[2]), n_b=int(args[3]), nnz_a=int(args[4]), nnz_b=int(args[5]),
                   base_a=base(args[6]), base_b=base(args[7]), op_a=opmap[args[8]], op_b=opmap[args[9]])
        if kind == "success":
            cfg["stage"] = "two-stage" if args[10] == "0" else "full"
            cfg["csr_a"] = not (len(args) > 11 and args[11] == "false")
            cfg["csr_b"] = not (len(args) > 12 and args[12] == "false")
        cfgs.append(cfg)
    out["sp2m_configs"] = dict(check="dense op(A)*op(B) within sqrt(eps) (sp2m_tests.cpp:501, 560-585)", cases=cfgs)

    # SpMV with extreme values, tests/unit_tests/mv_tests.cpp:1858-2100 (real types; the reference computes its expectation with
    # a plain loop over the same data, so the fixture holds the systems, where the special operands are planted and the list
    # of configurations; the tests evaluate the dense expression with IEEE semantics and match NaN / Inf / finite classes)
    S5 = dict(src="tests/unit_tests/common_data_utils.h:3897-4075", n=5, ptr=[0, 4, 6, 10, 13, 17],
              ind=[0, 2, 3, 4, 1, 4, 0, 2, 3, 4, 0, 2, 3, 0, 1, 2, 4],
              val=[211, 2.5, 1, 0.5, 271, 2, 2.5, 311, 1.2, 3, 1, 1.2, 287, 0.5, 2, 3, 251], x=[1, 2, 3, 4, 5], y0=[10.0] * 5,
              alpha=2.0, beta=2.0)
    NS5 = dict(src="tests/unit_tests/common_data_utils.h:4236-4330", n=5, ptr=[0, 4, 6, 7, 10, 14],
               ind=[0, 2, 3, 4, 1, 4, 4, 0, 2, 3, 0, 1, 2, 4], val=[211, 2.5, 1, 0.5, 271, 2, 3, 1, 1.2, 287, 0.5, 2, 3, 251],
               x=[1, 2, 3, 4, 5], y0=[0.0] * 5, alpha=1.0, beta=1.0)
    cfg = []
    msrc = open(os.path.join(REF, "mv_tests.cpp")).read()
    for mm in re.finditer(r"ADD_TEST\(\s*(EXT_\w+),\s*(\w+),\s*(\w+),\s*(\w+),\s*(\w+),\s*(\w+),\s*(\w+)\)", msrc):
        sysid, mtype, fmode, tr, op1, op2, rng = mm.groups()
        if sysid.startswith("EXT_H5"):
            continue  # hermitian data: complex types only
        mtype = "SYM" if mtype == "HERMIT" else mtype
        lower = fmode in ("LT", "SLT")
        if sysid.startswith("EXT_G5"):
            off = 9 if tr == "NT" else 15
        elif sysid.startswith("EXT_S5"):
            off = 15 if lower else 9
        else:
            off = 6
        cfg.append(dict(line=msrc[:mm.start()].count("\n") + 1, system="NS5" if sysid == "EXT_NSYMM_5" else "S5",
                        beta_zero=sysid.endswith("_B0"), type={"GEN": "general", "SYM": "symmetric", "TRIANG": "triangular"}[mtype],
                        fill="lower" if lower else "upper", diag="zero" if fmode in ("SLT", "SUT") else "non_unit",
                        op="n" if tr == "NT" else "t", val_offset=off, x_offset=4, op1=op1, op2=op2, range=rng,
                        x2_follows=(sysid.startswith("EXT_S5") and mtype == "SYM" and op2 == "ET_ZERO")))
    out["mv_extreme"] = dict(src="tests/unit_tests/mv_tests.cpp:1858-2100", systems=dict(S5=S5, NS5=NS5), configs=cfg,
                             operands="ET_NAN nan, ET_INF inf, ET_NUM unchanged, ET_ZERO 0; overflow / underflow: "
                                      "operands whose product stays inside (FLOW_EDGE_WITHIN) or leaves (FLOW_OUTOF) the range")
    # empty rows after optimize, tests/unit_tests/mv_tests.cpp:1319-1356: mv hint for op = none, product with op = transpose
    out["mv_empty_rows"] = dict(src="tests/unit_tests/mv_tests.cpp:1319-1356", m=5, n=5, ptr=[0, 0, 0, 1, 1, 1], ind=[2], val=[1.0],
                                x=[1.0, 2.0, 3.0, 4.0, 5.0], alpha=1.0, beta=0.0, hint_op="n", op="t", y_exp=[0, 0, 3, 0, 0])

    path = os.path.join(HERE, "reference_kats_r3.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes;", len(out["csrmm_runs"]), "csrmm runs,", len(cfgs), "sp2m configs")


if __name__ == "__main__":
    main()
