#!/usr/bin/env python3
"""bench.py -- CSR SpMV fp64 (aoclsparse_dmv through the C ABI) on N MI355X, one process per GPU.

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W`; for N > 1 the
driver starts it with torch.distributed.run, one rank per GPU; started WITHOUT torch.distributed.run (no WORLD_SIZE in
the environment) and N > 1 it starts the N ranks itself as a child process (self_launch below) and relays rank 0's line.

OUTPUT: the LAST stdout line is one short (<= 4 KB) strict-JSON record -- the reference harness prints one short statistics
line per run too (tests/common/aoclsparse_stats.cpp:266-340) -- carrying the headline, `roofline`, `cpu_baseline`, `parity`
and ONE number per leg (`legs`).  The full report (every leg with its quartiles, per-case rooflines and parity verdicts) goes
to the file named by --record (default bench_legs.json next to this file), not to stdout.

  step      = one aoclsparse_dmv (alpha=1, beta=0) over the whole matrix, x and y resident in HBM.
  workload  = 5-point Laplacian, BASELINE.json configs[1] scaled from its 100x100 grid (795 KB,
              launch-latency bound: reported in the "l100" object) to a 4096x4096 grid (m=16.8 M,
              nnz=83.9 M, 1.34 GB of algorithmic bytes > the 256 MiB Infinity Cache), as BASELINE.md
              section 3 row 2 prescribes for the roofline claim.
  value     = whole-job GFLOP/s = N * K * 2*nnz / max-over-ranks(time)   (SpMV does not shard: N
              independent replicas, "scaling": "weak"; DESIGN.md section "multi-GPU").
  roofline  = algorithmic bytes of one launch / average kernel time (hipEvents on the library's
              stream around the K back-to-back launches) against 8 TB/s HBM3E; "stats" holds the
              min / quartiles / max of the K per-step device times (the reference harness's statistics,
              tests/include/aoclsparse_stats.hpp:41-129).
  cpu_baseline = the CPU oracle (oracle/, a port of the reference's kernels and dispatch rule) on this box's
              host cores: one thread and all physical cores, first-touch arrays, threads bound to cores.
  legs      = the other BASELINE.json configs, one object each, every one with its own `roofline` object and a
              parity verdict against the oracle:
                l100                 configs[1] literal (10k x 10k), launch-latency bound
                dcsrmv_csr_adaptive  raw aoclsparse_dcsrmv on device-resident CSR arrays (CSR-Adaptive kernel)
                mix                  configs[2]: the four SuiteSparse matrices (real .mtx from $MATRIX_DIR, else the
                                     seeded stand-ins of tools/standins.py), kernel chosen by aoclsparse_optimize
                csrmm                configs[3] on ONE GPU: both layouts, 256 columns and the 32-column slab one of
                                     eight ranks owns
                csrmm_sharded        configs[3] over the N ranks (aocl-sparse_amd/sharded.py): A broadcast (RCCL), B/C
                                     column slabs, efficiency T1 / (N * TN), optional C all-gather
                trsv                 configs[4]: unit-lower ILU(0) factor of the shell-like matrix (and of its unstructured
                                     variant), automatic kid and the pinned KT orders (kid 1 / 3)
                sp2m                 SURVEY 8 a15: aoclsparse_sp2m(A, A) on the 1000^2 Laplacian and a 100,000-row shell mesh, host
                                     arrays in and out, against the CPU port of the reference's two-stage Gustavson
                spmv_row_sharded     SURVEY 8e "next": x <- A x iterated with A split by rows over the N ranks, one all-gather of
                                     the y slices per iteration (RCCL with nccl)
                sp2m_row_sharded     SURVEY 8e: C = A * A with A's rows split over the N ranks (independent slices, no data-path exchange)
                inlib_multi          configs[3] from ONE process: aoclsparse_mi355_dcsrmm_multi_slabs over every visible GPU
                                     (tools/multi_check.py as a child process)
              (csrmm also holds the pinned-kid cases; l100.c_caller is the per-call cost seen by a C program, tools/l100_probe.hip;
               --grid 10000 runs the 10,000 x 10,000-grid reading of configs[1]: m = 1e8, 8 GB per product)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# multi-process GPU work on this pool needs dmabuf IPC (RCCL fails with hipIpcGetMemHandle otherwise)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# CPU baseline: OpenMP threads bound to cores, close placement -- must be in the
# environment before the oracle's OpenMP runtime starts
os.environ.setdefault("OMP_PROC_BIND", "close")
os.environ.setdefault("OMP_PLACES", "cores")

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
FP64_MFMA_PEAK_TFLOPS = 78.6  # dense fp64 matrix peak of the part (= its fp64 vector rate: 256 CUs x 128 flop/clk x 2.4 GHz)
FP64_MFMA_SUSTAINED_TFLOPS = 47.7  # back-to-back v_mfma_f64_16x16x4_f64, measured (profiles/r4/mfma_f64_peak.jsonl)


# ---- pure helpers (unit-tested on CPU with gloo, tests/test_bench_dist_cpu.py) -------------------
def spmv_bytes(m, n, nnz, beta_nonzero=False):
    """Reference byte model, tests/include/aoclsparse_gbyte.hpp:39-45."""
    return (m + 1 + nnz) * 4 + (m + n + nnz) * 8 + (8 * m if beta_nonzero else 0)


def csrmm_bytes(m, k, nnz, ncols, beta_nonzero=False):
    """Dense-correct csrmm byte count."""
    return (m + 1 + nnz) * 4 + nnz * 8 + 8 * ncols * (k + m * (2 if beta_nonzero else 1))


def trsv_bytes(m, nnz_tri):
    """tests/include/aoclsparse_gbyte.hpp:67-71 (one triangle)."""
    return (m + 1 + nnz_tri) * 4 + (2 * m + nnz_tri) * 8


def column_shard(ncols, world, rank):
    """[j0, j1) of rank `rank`: the reference's per-thread column split of csrmm
    (library/src/level3/aoclsparse_csrmm_kt.cpp:68-82) with ranks in place of threads; the same rule as the
    library's aoclsparse_mi355_column_shard (kept in python too so that it is testable without the .so)."""
    def edge(t):
        e = ncols * t // world
        if e % 4:
            e += 4 - e % 4
        return min(e, ncols)
    return edge(rank), edge(rank + 1)


def reduce_scalar(value, op, dist=None, device="cpu"):
    """max / sum of a python float over all ranks (identity when not distributed)."""
    if dist is None or not dist.is_initialized():
        return float(value)
    import torch

    dev = "cpu" if dist.get_backend() == "gloo" else device
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM)
    return float(t.item())


def job_throughput(units_this_rank, seconds_this_rank, dist=None, device="cpu"):
    """(sum of units over ranks) / (max of time over ranks) -- the whole-job figure."""
    tmax = reduce_scalar(seconds_this_rank, "max", dist, device)
    usum = reduce_scalar(units_this_rank, "sum", dist, device)
    return usum / tmax, tmax


def quartiles(ms):
    """min / q1 / median / q3 / max of per-iteration times (tests/include/aoclsparse_stats.hpp:41-129)."""
    import numpy as np

    a = np.sort(np.asarray(ms, dtype=np.float64))
    if len(a) == 0:
        return None
    q = lambda f: round(float(np.quantile(a, f)), 6)
    return {"min": round(float(a[0]), 6), "q1": q(0.25), "median": q(0.5), "q3": q(0.75),
            "max": round(float(a[-1]), 6), "n": int(len(a))}


def roofline(abytes, ms, traffic=None, **extra):
    gbs = abytes / (ms * 1e-3) / 1e9
    out = {"bound": "hbm", "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": traffic, "algorithmic_bytes_per_launch": int(abytes),
           "kernel_ms": round(ms, 6)}
    if traffic:
        # bytes the kernel really moved per launch (PMC) and the rate it moved them at: a format that stores less than the
        # CSR model counts (SELL-64 with shared column lists) has achieved > traffic_gbs, and achieved may exceed what a copy does
        # (on float even the 8 TB/s pin rate) -- frac_traffic is the fraction of the roof in REAL bytes, the twin of frac
        out["traffic_gbs"] = round(traffic / (ms * 1e-3) / 1e9, 2)
        out["frac_traffic"] = round(traffic / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        out["traffic_over_algorithmic"] = round(traffic / abytes, 4)
    out.update(extra)
    return out


_FLOOR_CACHE = {}


def latency_floor(workgroups, footprint_bytes, entries, achieved_us, lanes=256):
    """`roofline_latency` of a leg whose working set stays in the L2s / Infinity Cache (8 TB/s of HBM is the wrong roof there):
    the measured floor of a kernel with the same launch shape that does the CSR kernels' minimum -- launch ramp, THREE dependent
    memory round trips per wavefront (block table -> column indices / values -> x[col]) over the same footprint, then a coalesced
    12-byte-per-entry stream of the same length (tools/latency_floor.hip, run as a child process) -- next to the achieved time.
    frac = floor / achieved.  None when the probe is not built or fails."""
    import subprocess

    probe = os.path.join(ROOT, "tools", "bin", "latency_floor")
    key = (int(workgroups), max(1, int(footprint_bytes) >> 20), int(entries), 128 if lanes == 128 else 256)
    if key not in _FLOOR_CACHE:
        res = None
        try:
            if os.path.exists(probe):
                r = subprocess.run([probe] + [str(k) for k in key], capture_output=True, text=True, timeout=120)
                line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
                if r.returncode == 0 and line:
                    res = json.loads(line[-1])
        except Exception:  # noqa: BLE001 -- an extra
            res = None
        _FLOOR_CACHE[key] = res
    res = _FLOOR_CACHE[key]
    if not res:
        return None
    floor = float(res["three_hops_then_stream_us"])
    # (the probe is a reference kernel of the same shape, not a proof of optimality: the product may come out a few % under it)
    return {"bound": "latency", "floor_us": round(floor, 3), "achieved_us": round(achieved_us, 3),
            "frac": round(min(1.0, floor / achieved_us), 4) if achieved_us > 0 else None, "unit": "us",
            "floor_over_achieved_raw": round(floor / achieved_us, 4) if achieved_us > 0 else None,
            "empty_kernel_us": res["empty_kernel_us"], "three_dependent_trips_us": res["three_hops_us"],
            "workgroups": key[0], "lanes_per_workgroup": key[3], "footprint_mb": key[1], "entries": key[2],
            "how": "tools/latency_floor.hip: launch + 3 dependent round trips per wavefront + a coalesced stream of the same "
                   "length, same number of workgroups; the working set fits the L2s / Infinity Cache, so HBM is not the bound"}


def cpu_info():
    """CPU model string, logical CPUs this process may use, physical cores among them."""
    model, cores = "unknown", set()
    try:
        phys = core = None
        allowed = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else None
        cpu = None
        with open("/proc/cpuinfo") as f:
            for line in f:
                k, _, v = line.partition(":")
                k, v = k.strip(), v.strip()
                if k == "processor":
                    cpu = int(v)
                elif k == "model name":
                    model = v
                elif k == "physical id":
                    phys = v
                elif k == "core id":
                    core = v
                    if allowed is None or cpu in allowed:
                        cores.add((phys, core))
    except OSError:
        pass
    logical = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return model, logical, (len(cores) or logical)


def timed_laps(pkg, fn, steps, warmup):
    """`steps` back-to-back calls of fn on the library's stream, one hipEvent between consecutive calls
    -> per-step device milliseconds (the events live on the stream the kernels are launched on)."""
    for _ in range(warmup):
        fn()
    pkg.lib().aoclsparse_mi355_synchronize()
    pkg.timer_mark()
    for _ in range(steps):
        fn()
        pkg.timer_mark()
    return pkg.timer_laps()


def timed_cold(pkg, fn, steps, flush, warmup=2):
    """`steps` calls of fn with the Infinity Cache FLUSHED before each one: `flush` (1 GiB of other data) is read by a device kernel
    between any two calls, so nothing of the operands is left in the 256 MiB cache from the call before (a FILL as the flush would
    leave dirty lines whose write-back runs into the product: tools/history/exp_cold.py).  One hipEvent on the library's stream
    on each side of every call (the flush runs on the null stream, which the library's blocking stream is ordered with)
    -> per-call device milliseconds: the HBM-only time of the call.  The model is the reference harness's byte accounting per cold
    call, tests/include/aoclsparse_gbyte.hpp:39-45."""
    for _ in range(warmup):
        flush.sum()
        fn()
    pkg.lib().aoclsparse_mi355_synchronize()
    pkg.timer_laps()  # (drop marks of earlier users)
    for _ in range(steps):
        flush.sum()
        pkg.timer_mark()
        fn()
        pkg.timer_mark()
    return pkg.timer_laps()[0::2]  # (the odd laps are the flushes)


def bound_of(working_set_bytes):
    """what a back-to-back loop over a working set of this size is bound by: the 8 x 4 MiB L2s keep <= 32 MiB (launch + dependent
    round trips: latency), the Infinity Cache <= 256 MiB, beyond that HBM (MI355X_MICROARCH.md, memory hierarchy)"""
    if working_set_bytes <= (32 << 20):
        return "latency"
    return "infinity_cache" if working_set_bytes <= (256 << 20) else "hbm"


COMPACT_LIMIT = 4096  # bytes: the driver keeps a bounded tail of stdout, the record must fit it with room to spare


def _short(text, n):
    text = "" if text is None else str(text)
    return text if len(text) <= n else text[: n - 3] + "..."


def _eff8(t1_ms, slab_ms):
    """projected 8-GPU compute efficiency of the column-sharded csrmm: T1 / (8 * T_slab)"""
    return round(t1_ms / (8.0 * slab_ms), 4) if t1_ms and slab_ms else None


def leg_numbers(full):
    """ONE number per leg out of the full report (None where the leg did not run)."""
    legs = full.get("legs") or {}
    n = {}
    l100 = full.get("l100") or {}
    if "us_per_call" in l100:
        n["l100_us"] = l100["us_per_call"]
        n["l100_latency_frac"] = (l100.get("roofline_latency") or {}).get("frac")
        # l100_us is the mean of 2,000 calls each bracketed by its own event pair, from Python; what a C caller pays per call
        # (tools/l100_probe.hip, next to an empty launch with the same arguments) and the per-call time inside a HIP graph:
        cc = (l100.get("c_caller") or {}).get("dmv_pointer_mode_device") or {}
        if "total_us" in cc:
            n["l100_c_caller_us"] = cc["total_us"]
        if "us_per_call_in_a_hip_graph_of_100" in l100:
            n["l100_graph_us"] = l100["us_per_call_in_a_hip_graph_of_100"]
    ca = legs.get("dcsrmv_csr_adaptive") or {}
    if "roofline" in ca:
        n["csr_adaptive_ms"] = ca["ms"]
        n["csr_adaptive_frac"] = ca["roofline"]["frac"]
        if ca["roofline"].get("traffic_over_algorithmic"):
            n["csr_adaptive_traffic_over_algorithmic"] = ca["roofline"]["traffic_over_algorithmic"]
    tw = legs.get("headline_twins") or {}
    if "ms" in (tw.get("cold_dmv") or {}):
        n["dmv_cold_ms"] = tw["cold_dmv"]["ms"]
    if "smv" in tw:
        n["smv_frac"] = tw["smv"]["roofline"]["frac"]
        n["smv_frac_traffic"] = tw["smv"]["roofline"].get("frac_traffic")
        n["smv_parity"] = tw["smv"]["bit_exact_first_2e20_rows"]
    if "host_pointer_dmv" in tw:
        n["host_ptr_dmv_ms"] = tw["host_pointer_dmv"]["ms_wall_per_call"]
    mix = legs.get("mix") or {}
    rows = mix.get("matrices") or []
    if rows:
        primary = [r["roofline"]["frac"] for r in rows if "," not in r["matrix"]]
        n["mix_frac_mean"] = round(sum(primary) / len(primary), 4) if primary else None
        n["mix_frac"] = {_short(r["matrix"].split(" (")[0], 28): r["roofline"]["frac"] for r in rows if "," not in r["matrix"]}
        if all("roofline_cold" in r for r in rows):
            # cache flushed before every product: the HBM fractions (mix_frac above is the back-to-back loop, an HBM fraction only
            # where mix_bound says "hbm")
            cold = [r["roofline_cold"]["frac"] for r in rows if "," not in r["matrix"]]
            n["mix_cold_frac_mean"] = round(sum(cold) / len(cold), 4) if cold else None
            n["mix_cold_frac"] = {_short(r["matrix"].split(" (")[0], 28): r["roofline_cold"]["frac"] for r in rows if "," not in r["matrix"]}
            n["mix_bound"] = {_short(r["matrix"].split(" (")[0], 28): r["bound_back_to_back"] for r in rows if "," not in r["matrix"]}
        lat = {_short(r["matrix"].split(" (")[0], 28): r["roofline_latency"]["frac"] for r in rows
               if r.get("roofline_latency") and "," not in r["matrix"]}
        if lat:
            n["mix_latency_frac"] = lat
        n["mix_parity"] = all(r.get("bit_exact_rows_below_tree_min", r.get("bit_exact_rows_within_tile")) and r["long_rows_within_bound"]
                              and (r.get("strict_mode") or {}).get("bit_exact_every_row", True) for r in rows)
    mm = legs.get("csrmm") or {}
    for lay, key in (("row-major", "row"), ("column-major", "col")):
        for mode, suffix in (("default", ""), ("opt-in", "_overwrite")):
            cs = [c for c in mm.get("cases", []) if c["layout"] == lay and c["mode"].startswith(mode)]
            fullc = [c for c in cs if c["what"].startswith("all")]
            slab = [c for c in cs if c["what"].startswith("one slab")]
            if fullc and slab:
                n["csrmm_%s%s_ms" % (key, suffix)] = fullc[0]["ms"]
                n["csrmm_%s%s_slab_ms" % (key, suffix)] = slab[0]["ms"]
                if "roofline_cold" in fullc[0] and "roofline_cold" in slab[0]:
                    # fractions of the HBM roof on the bytes the mode must move, cache flushed before every product; ONE projected
                    # 8-GPU efficiency per cell from the cold times, T1 / (8 T_slab) (it RISES when the full-width kernel gets slower:
                    # the slab fraction is the figure of merit)
                    n["csrmm_%s%s_frac" % (key, suffix)] = fullc[0]["roofline_cold"]["frac"]
                    n["csrmm_%s%s_slab_frac" % (key, suffix)] = slab[0]["roofline_cold"]["frac"]
                    if not suffix:
                        n["csrmm_%s_eff8_cold" % key] = _eff8(fullc[0]["cold_ms"], slab[0]["cold_ms"])
                else:
                    n["csrmm_%s%s_eff8" % (key, suffix)] = _eff8(fullc[0]["ms"], slab[0]["ms"])
    if mm.get("cases"):
        n["csrmm_parity"] = all(c.get("bit_exact_4_columns", True) and c.get("bit_exact_8_columns_vs_kt_oracle", True)
                                for c in mm["cases"])
    if (mm.get("blocked") or {}).get("cases"):
        b = mm["blocked"]
        n["csrmm_blocked_mfma_ms"] = b.get("mfma_ms")
        n["csrmm_blocked_mfma_col_ms"] = b.get("mfma_col_ms")
        n["csrmm_blocked_mfma_tflops"] = b["cases"][0]["mfma"].get("tflops")
        n["csrmm_blocked_mfma_overwrite_tflops"] = (b["cases"][0]["mfma"].get("overwrite") or {}).get("tflops")
        n["csrmm_blocked_parity"] = b.get("parity_ok")
    tr = legs.get("trsv") or {}
    if tr.get("schedules"):
        n["trsv_ms"] = tr["schedules"][0]["ms"]
        n["trsv_parity"] = all(s["bit_exact_vs_cpu"] for s in tr["schedules"])
        if tr.get("plan"):
            n["trsv_schedule"] = tr["plan"]["automatic_schedule"]
            n["trsv_us_per_block_level"] = round(tr["schedules"][0]["ms"] * 1e3 / max(tr["plan"]["block_levels"], 1), 4)
        if "unstructured_variant" in tr and tr["unstructured_variant"].get("schedules"):
            n["trsv_unstructured_ms"] = tr["unstructured_variant"]["schedules"][0]["ms"]
            n["trsv_unstructured_us_per_level"] = tr["unstructured_variant"]["schedules"][0].get("us_per_level")
            n["trsv_unstructured_schedule"] = (tr["unstructured_variant"].get("plan") or {}).get("automatic_schedule")
    sp2 = legs.get("sp2m") or {}
    if sp2.get("cases"):
        n["sp2m_ms"] = sp2["cases"][0]["ms"]
        n["sp2m_parity"] = all(c["bit_exact"] for c in sp2["cases"])
    for lay in ("col", "row", "bell"):
        sh = full.get("csrmm_sharded_" + lay) or {}
        if "tg_ms_device_median_max_over_ranks" in sh and sh.get("world", 1) > 1:  # (one rank: the csrmm leg above says it all)
            n["csrmm_sharded_" + lay] = {"world": sh["world"], "cols_per_rank": sh["cols_per_rank"],
                                         "tg_ms": sh["tg_ms_device_median_max_over_ranks"], "t1_ms": sh.get("t1_ms"),
                                         "slab_ms_per_rank": sh.get("slab_ms_per_rank"),
                                         "efficiency": sh.get("efficiency"), "a_broadcast_ms": sh.get("a_broadcast_ms"),
                                         "c_allgather_ms": sh.get("c_allgather_ms"),
                                         "parity": (sh.get("parity") or {}).get("bit_exact")}
    s2 = full.get("sp2m_row_sharded") or {}
    if "product_ms_median_max_over_ranks" in s2 and s2.get("world", 1) > 1:
        n["sp2m_row_sharded"] = {"world": s2["world"], "product_ms": s2["product_ms_median_max_over_ranks"], "nnz_c": s2["nnz_c"],
                                 "parity": (s2.get("parity") or {}).get("bit_exact")}
    sp = full.get("spmv_row_sharded") or {}
    if "product_ms_median_max_over_ranks" in sp and sp.get("world", 1) > 1:
        n["spmv_row_sharded"] = {"world": sp["world"], "m": sp["m"], "product_ms": sp["product_ms_median_max_over_ranks"],
                                 "allgather_ms": sp["allgather_ms_median_max_over_ranks"],
                                 "shard_frac": (sp.get("roofline_shard") or {}).get("frac"),
                                 "parity": (sp.get("parity") or {}).get("bit_exact")}
    im = legs.get("inlib_multi") or {}
    if im:
        n["inlib_multi"] = {k: im.get(k) for k in ("devices", "same_device", "slabs_bit_exact", "efficiency_wall") if k in im}
    errors = [k for k, v in list(legs.items()) + [(k, full.get(k)) for k in ("l100", "spmv_row_sharded", "sp2m_row_sharded",
                                                                             "csrmm_sharded_col", "csrmm_sharded_row", "csrmm_sharded_bell")]
              if isinstance(v, dict) and "error" in v]
    if errors:
        n["errors"] = errors
    return n


def compact_record(full, record_path=None):
    """The driver-facing record: strict JSON, <= COMPACT_LIMIT bytes, the contract's keys + roofline + cpu_baseline + one
    number per leg.  Everything else stays in the full report (`record_path`)."""
    c = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                  "scaling", "vs_baseline", "dtype", "data")}
    if "value_back_to_back" in full:
        c["value_back_to_back"] = full["value_back_to_back"]
    cfg = full.get("config") or {}
    c["config"] = {"workload": _short(cfg.get("workload"), 200), "kernel": _short(cfg.get("kernel"), 160),
                   "parallelism": cfg.get("parallelism"), "device": _short(cfg.get("device"), 40),
                   "communicator": cfg.get("communicator")}
    rf = full.get("roofline") or {}
    c["roofline"] = {k: rf.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "frac_traffic", "traffic",
                                            "algorithmic_bytes_per_launch", "kernel_ms", "traffic_source")}
    for k in ("frac_back_to_back", "kernel_ms_back_to_back"):
        if k in rf:
            c["roofline"][k] = rf[k]
    if rf.get("note"):
        c["roofline"]["note"] = _short(rf["note"], 180)
    if c["roofline"].get("frac_traffic") is None and rf.get("traffic") and rf.get("kernel_ms") and rf.get("peak"):
        c["roofline"]["frac_traffic"] = round(rf["traffic"] / (rf["kernel_ms"] * 1e-3) / 1e9 / rf["peak"], 4)  # (reports of rounds 1-4)
    cb = full.get("cpu_baseline")
    if isinstance(cb, dict) and "value" in cb:
        c["cpu_baseline"] = {"value": cb["value"], "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                             "cpu_model": _short(cb.get("cpu_model"), 60), "one_thread_value": (cb.get("one_thread") or {}).get("gflops"),
                             "bit_exact_vs_gpu": cb.get("bit_exact_vs_gpu"), "sample": _short(cb.get("sample"), 160)}
    else:
        c["cpu_baseline"] = cb  # None on N > 1 runs / {"error": ...}
    c["parity"] = full.get("parity")
    c["legs"] = leg_numbers(full)
    if record_path:
        c["full_report"] = record_path
    line = json.dumps(c, allow_nan=False, separators=(",", ":"))
    if len(line) > COMPACT_LIMIT:  # never let prose cost the record: drop the optional strings, then the per-matrix maps
        for k in ("sample",):
            if isinstance(c.get("cpu_baseline"), dict):
                c["cpu_baseline"].pop(k, None)
        c["config"]["workload"] = _short(c["config"]["workload"], 80)
        c["config"]["kernel"] = _short(c["config"]["kernel"], 60)
        line = json.dumps(c, allow_nan=False, separators=(",", ":"))
    if len(line) > COMPACT_LIMIT:
        for k in ("mix_frac", "mix_latency_frac", "inlib_multi"):
            c["legs"].pop(k, None)
        line = json.dumps(c, allow_nan=False, separators=(",", ":"))
    if len(line) > COMPACT_LIMIT:  # the sharded objects next, then every leg: the headline, roofline and cpu_baseline always fit
        c["legs"] = {k: v for k, v in c["legs"].items() if not isinstance(v, dict)}
        line = json.dumps(c, allow_nan=False, separators=(",", ":"))
    if len(line) > COMPACT_LIMIT:
        c["legs"] = {"dropped": "see full_report"}
        line = json.dumps(c, allow_nan=False, separators=(",", ":"))
    return line


def _sanitize(o):
    """NaN / Inf -> None so that the records are STRICT JSON"""
    if isinstance(o, float):
        return o if o == o and o not in (float("inf"), float("-inf")) else None
    if isinstance(o, dict):
        return {str(k): _sanitize(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_sanitize(v) for v in o]
    return o


def self_launch(argv, gpus):
    """`python bench.py --gpus N` with N > 1 and no torch.distributed.run around it: start the N ranks as a CHILD process (this
    parent has made no GPU call and makes none -- it never even imports torch), pass the child's stdout through, print rank 0's
    record again as the LAST line and return the child's exit code (non-zero on any rank's failure)."""
    import socket
    import subprocess

    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", AOCLSPARSE_BENCH_SELF_LAUNCHED="1")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // gpus)))
    # The rendezvous port is found by binding port 0 and closing the socket again: another process may take it before the child
    # binds it (two benches or tests on one box).  A child that dies within seconds without having printed anything of the
    # record is started once more on a fresh port.
    for attempt in (0, 1):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr",
               "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
        t0 = time.perf_counter()
        proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, cwd=ROOT)
        record = None
        for line in proc.stdout:
            line = line.rstrip("\n")
            if line.startswith("{") and '"metric"' in line:
                record = line  # held back: it must be the last line
            else:
                print(line, flush=True)
        rc = proc.wait()
        if rc == 0 or record is not None or attempt == 1 or time.perf_counter() - t0 > 45.0:
            break
        print("bench.py: the %d-rank child exited with %d after %.1f s without a record (rendezvous port %d taken?): one more "
              "try on a fresh port" % (gpus, rc, time.perf_counter() - t0, port), file=sys.stderr)
    if record is not None:
        print(record, flush=True)
    if rc == 0 and record is None:
        print("bench.py: the %d-rank child printed no record" % gpus, file=sys.stderr)
        rc = 1
    return rc



# ---- the benchmark ------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--grid", type=int, default=4096, help="Laplacian grid edge of the headline workload")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: CPU tensors on the wire, ranks may share one GPU (control-flow tests on a 1-GPU box)")
    ap.add_argument("--legs", default="all",
                    help="comma list of l100,dcsrmv_csr_adaptive,headline_twins,mix,csrmm,csrmm_sharded,spmv_row_sharded,sp2m_row_sharded,trsv,sp2m,cpu,inlib_multi (or all / none)")
    ap.add_argument("--mm-grid", type=int, default=1000, help="csrmm: A = Laplacian on grid^2")
    ap.add_argument("--mm-cols", type=int, default=256)
    ap.add_argument("--shard-grid", type=int, default=0,
                    help="grid of the row-sharded SpMV iteration leg; 0 = 4096*sqrt(N) (every rank keeps the headline's 16.8 M rows = "
                         "1.34 GB of matrix, well past the 256 MiB Infinity Cache, whatever N is); 2048 on one rank")
    ap.add_argument("--bell-nodes", type=int, default=40,
                    help="sharded csrmm on the block-dense stand-in (blocked-ELL MFMA path): nodes per edge of its node grid, 16 "
                         "unknowns each (40 -> 1,024,000 rows: configs[3]'s 1M x 1M); 0 skips it")
    ap.add_argument("--sp2m-grid", type=int, default=1000, help="sp2m_row_sharded: A = 5-pt Laplacian on grid^2 (rows split over the ranks)")
    ap.add_argument("--shard-own-rows", action="store_true",
                    help="row-sharded SpMV leg: every rank builds its own rows even with an explicit --shard-grid (the default for "
                         "N > 1 without --shard-grid; lets a small test take the path the multi-GPU run takes)")
    ap.add_argument("--mm-layout", default="both", choices=["col", "row", "both"],
                    help="layout(s) of the sharded csrmm leg: column-major slabs are the contiguous ones (SURVEY.md 8e)")
    ap.add_argument("--dry-launch", action="store_true",
                    help="launch check only: the ranks rendezvous (gloo), sum their ranks and rank 0 prints a record with "
                         "metric 'launch-check'; no GPU is touched and nothing is measured (CPU-tier test of the launch convention)")
    ap.add_argument("--record", default=os.path.join(ROOT, "bench_legs.json"),
                    help="file that receives the FULL report (stdout only gets the short record); '' = do not write it")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="CPU-baseline budget per thread count")
    ap.add_argument("--small", action="store_true", help="mix / trsv legs on the two small matrices only")
    ap.add_argument("--cold-only", action="store_true",
                    help="skip timed region 2 (the back-to-back products): every launch of the headline kernel in the run is then a "
                         "cold one, so a rocprofv3 --kernel-trace --stats average of that kernel is the cold average")
    args = ap.parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # not under torch.distributed.run: start the ranks ourselves (as a child; nothing here has touched the GPU)
        sys.exit(self_launch(sys.argv[1:], args.gpus))
    all_legs = ["l100", "dcsrmv_csr_adaptive", "headline_twins", "mix", "csrmm", "csrmm_sharded", "spmv_row_sharded", "sp2m_row_sharded", "trsv", "sp2m", "cpu",
                "inlib_multi"]
    legs = set(all_legs) if args.legs == "all" else set(x for x in args.legs.split(",") if x and x != "none")
    assert legs <= set(all_legs), "unknown leg in --legs: %s" % sorted(legs - set(all_legs))

    # host description BEFORE any OpenMP runtime is loaded (torch brings libgomp): with OMP_PROC_BIND set, libgomp pins
    # the thread that loads it to its first place, after which sched_getaffinity() reports that one core only
    cpu_model, cpu_logical, cpu_physical = cpu_info()
    main_affinity = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else None

    def unpin():
        """give the launching thread its original CPU mask back (after the OpenMP runtime start and oracle calls)"""
        if main_affinity is not None:
            try:
                os.sched_setaffinity(0, main_affinity)
            except OSError:
                pass

    import numpy as np
    import torch
    import torch.distributed as dist

    import __graft_entry__ as entry

    unpin()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # fail loudly, never fall back: --gpus N is N ranks (one process per GPU over RCCL); a launcher that set WORLD_SIZE to
    # something else is a mistake of the launch, not something to paper over
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d needs WORLD_SIZE=%d (launch with python -m torch.distributed.run --nproc-per-node %d ...);"
                 " found WORLD_SIZE=%d.  No single-process fallback." % (args.gpus, args.gpus, args.gpus, world))
    if args.dry_launch:
        # the launch convention alone: rendezvous, one all-reduce, rank 0's record as the last line.  NOT a measurement.
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        ssum = float(rank)
        if world > 1 or "RANK" in os.environ:
            dist.init_process_group("gloo", rank=rank, world_size=world)
            ssum = reduce_scalar(float(rank), "sum", dist)
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"metric": "launch-check", "value": None, "n_gpus": world, "rank_sum": ssum,
                              "self_launched": os.environ.get("AOCLSPARSE_BENCH_SELF_LAUNCHED") == "1"}), flush=True)
        return
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the product has no CPU path")
    ndev = torch.cuda.device_count()
    if args.backend == "nccl" and world > ndev:
        sys.exit("bench.py: %d ranks but %d visible GPU(s): RCCL needs one GPU per rank (use --backend gloo to share one GPU "
                 "for control-flow tests)" % (world, ndev))
    dev_index = local_rank % ndev if args.backend == "gloo" else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    use_dist = world > 1 or "RANK" in os.environ  # under torch.distributed.run: a process group even for 1 rank
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    D = dist if use_dist else None
    if use_dist:
        # the first real multi-GPU run checks itself: the communicator every collective below uses spans exactly --gpus ranks, and
        # (RCCL) says which version it is
        cinfo = {"backend": dist.get_backend(), "world": dist.get_world_size()}
        assert cinfo["world"] == args.gpus == world, "communicator spans %d ranks, --gpus %d" % (cinfo["world"], args.gpus)
        if cinfo["backend"] == "nccl":
            assert torch.cuda.nccl.version(), "RCCL reports no version"

    pkg = entry.load_package()
    import aocl_sparse_amd.sharded as sharded

    L = pkg.lib()
    st, dev_id, cus, dev_name = pkg.device_info()
    assert st == 0, "HIP runtime failed to initialise"
    L.aoclsparse_mi355_set_pointer_mode(pkg.PTR_DEVICE)  # every vector below lives in HBM

    def barrier():
        sharded.barrier(D, torch)

    # ---------------- headline: scaled 5-pt Laplacian dmv ----------------
    g = args.grid
    csr = entry.laplace5(g) if rank == 0 or not use_dist else None
    if use_dist and world > 1:
        # built once (rank 0) and broadcast: RCCL GPU-to-GPU with nccl, CPU tensors with gloo
        csr5, lap_bcast_ms = sharded.broadcast_csr(D, torch, device, rank, (csr[0], csr[0]) + tuple(csr[1:]) if rank == 0 else None)
        m, _, row_ptr, col_ind, val = csr5
    else:
        m, row_ptr, col_ind, val = entry.laplace5(g) if csr is None else csr
        lap_bcast_ms = 0.0
    nnz = int(len(val))
    A = pkg.Matrix(0, m, m, row_ptr, col_ind, val)
    assert A.status == 0
    descr = pkg.Descr()
    assert L.aoclsparse_set_mv_hint(A.h, pkg.OP_NONE, descr.h, args.steps + args.warmup) == 0
    assert L.aoclsparse_optimize(A.h) == 0  # uploads CSR to HBM + builds SELL-64 / the row-block plan
    info = A.spmv_info()
    xh = np.sin(0.01 * np.arange(m))
    x = torch.from_numpy(xh).to(device)
    y = torch.zeros(m, dtype=torch.float64, device=device)

    def step():
        s = pkg.dmv(pkg.OP_NONE, 1.0, A, descr, x, 0.0, y)
        assert s == 0, pkg.STATUS[s]

    # Timed region 1 (the headline: `value`, `ms_per_step`, `roofline`): K products, each from a COLD Infinity Cache -- 1 GiB of
    # other data is read between any two products (timed_cold's flush), one hipEvent on the launch stream on each side of every
    # product.  A step's time is its product's device time; the flush between steps is not part of any step.  Back-to-back
    # products of one handle keep the end of their (alternating) sweep in the 256 MiB cache: that figure is region 2 below and
    # carries its own names (value_back_to_back, roofline.frac_back_to_back) -- a fraction of the HBM roof has to be HBM traffic.
    flush = torch.ones(1 << 28, dtype=torch.float32, device=device)
    for _ in range(args.warmup):
        flush.sum()
        step()
    L.aoclsparse_mi355_synchronize()
    pkg.timer_laps()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        flush.sum()
        pkg.timer_mark()
        step()
        pkg.timer_mark()
    all_laps = pkg.timer_laps()  # drains the stream
    barrier()
    wall_cold = time.perf_counter() - t0
    laps = all_laps[0::2]  # per-product device ms (the odd laps are the flushes)
    assert len(laps) == args.steps
    elapsed = float(sum(laps)) * 1e-3  # seconds of the K cold products on this rank

    # Timed region 2: the same K products back to back (what an iterative solver that does nothing else between products sees)
    b2b = None
    if not args.cold_only:
        for _ in range(args.warmup):
            step()
        barrier()
        t1 = time.perf_counter()
        pkg.timer_mark()
        for _ in range(args.steps):
            step()
            pkg.timer_mark()
        laps_b2b = pkg.timer_laps()
        barrier()
        elapsed_b2b = time.perf_counter() - t1
        b2b = (elapsed_b2b, laps_b2b)
    del flush

    flops = 2.0 * nnz
    abytes = spmv_bytes(m, m, nnz)
    gflops, tmax = job_throughput(args.steps * flops / 1e9, elapsed, D, device)
    kernel_ms = float(sum(laps)) / max(len(laps), 1)  # average launch duration over the timed region (cold products)
    stats = quartiles(laps)
    if b2b is not None:
        gflops_b2b, tmax_b2b = job_throughput(args.steps * flops / 1e9, b2b[0], D, device)
        kernel_ms_b2b = float(sum(b2b[1])) / max(len(b2b[1]), 1)

    traffic, traffic_src = None, None
    try:  # PMC counters cannot be read from inside the run: use the committed rocprofv3 measurement
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            pmc = json.load(f)
        if pmc.get("grid") == g and pmc.get("kernel_id", 1) == info.kernel:
            traffic, traffic_src = pmc["traffic_bytes_per_launch"], "profiles/pmc_traffic.json"
    except (OSError, ValueError, KeyError):
        pass

    out = {
        "metric": "CSR SpMV fp64 GFLOP/s + achieved-HBM-GB/s %roofline",
        "value": round(gflops, 3),
        "unit": "GFLOP/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(tmax / args.steps * 1e3, 6),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": "aoclsparse_dmv, 5-pt Laplacian grid %dx%d (m=%d, nnz=%d), alpha=1 beta=0; "
                        "BASELINE configs[1] scaled past the 256 MiB Infinity Cache" % (g, g, m, nnz),
            # (the first 160 characters go into the short record: what a reader of `frac` must know comes first)
            "kernel": ("SELL-64, %d slices%s; every timed product starts from a flushed Infinity Cache (1 GiB read between "
                       "products): HBM-only; %.3f cells per nnz, built by "
                       "aoclsparse_optimize for the mv hint, order %d (reference ref_csrmv_gn order)%s"
                       % (info.sell_slices, ", shared column lists" if info.kernel == 4 else "", info.stored_cells / max(nnz, 1),
                          info.order, ": ONE column list per run of rows that repeat it (as it is or shifted by one: a stencil's "
                          "rows) instead of one per row" if info.kernel == 4 else ""))
                      if info.kernel in (3, 4) else
                      ("csr-adaptive stream, order %d (reference ref_csrmv_gn order), %d row blocks; every timed product starts from a "
                       "flushed Infinity Cache" % (info.order, info.row_blocks)),
            "parallelism": "replicas x%d" % world,
            "device": dev_name,
            "backend": args.backend if use_dist else "none",
            "communicator": sharded.communicator_info(D, torch),
        },
        "roofline": roofline(abytes, kernel_ms, traffic, traffic_source=traffic_src,
                             stored_format_bytes_per_launch=(info.stored_cells * 12 + 16 * m) if info.kernel == 3
                             else (None if info.kernel == 4 else abytes),  # kernel 4: see roofline.traffic (PMC)
                             achieved_at_median=round(abytes / (stats["median"] * 1e-3) / 1e9, 2),
                             note=("cold products (cache flushed before each): frac = CSR-model bytes / time / peak, HBM only; the format "
                                   "moves `traffic` bytes (PMC): frac_traffic; back-to-back figure: frac_back_to_back")
                             if info.kernel in (3, 4) else None),
        "stats": dict(stats, unit="ms per cold step (device, one hipEvent on each side of every product)"),
        "timing": {"step": "one aoclsparse_dmv from a flushed Infinity Cache; ms_per_step = max over ranks of the mean device time "
                           "of the K products (hipEvents on the launch stream); the 1 GiB flush read between steps belongs to no step",
                   "wall_ms_per_step_including_the_flush": round(wall_cold / args.steps * 1e3, 6)},
    }
    if b2b is not None:
        # the cache-assisted twin, named as such: K products back to back, wall clock between barriers (rounds 1-5's headline)
        out["value_back_to_back"] = round(gflops_b2b, 3)
        out["ms_per_step_back_to_back"] = round(tmax_b2b / args.steps * 1e3, 6)
        out["roofline"]["frac_back_to_back"] = round(abytes / (kernel_ms_b2b * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        out["roofline"]["kernel_ms_back_to_back"] = round(kernel_ms_b2b, 6)
        out["roofline"]["bound_back_to_back"] = "hbm + infinity_cache (alternating sweeps: the end of one product is the start of the next)"
        out["stats_back_to_back"] = dict(quartiles(b2b[1]), unit="ms per step (device, hipEvent between consecutive launches)")
    if lap_bcast_ms:
        out["config"]["matrix_broadcast_ms"] = round(lap_bcast_ms, 2)

    legs_out = {}

    def run_leg(name, fn, collective=False):
        """collective legs run on every rank; the others on rank 0 of a 1-GPU run only (they are per-GPU figures and
        would only lengthen the scaling runs)."""
        if name not in legs:
            return
        if not collective and (rank != 0 or world > 1):
            return
        t = time.perf_counter()
        try:
            res = fn()
        except Exception as e:  # a supplementary leg must never cost the headline line
            res = {"error": "%s: %s" % (type(e).__name__, e)}
        if isinstance(res, dict):
            res["leg_seconds"] = round(time.perf_counter() - t, 2)
        if rank == 0:
            legs_out[name] = res

    # ---------------- collectives first: the column-sharded csrmm (every rank) ----------------
    def leg_csrmm_sharded(layout, matrix="laplace"):
        csr_mm = None
        if rank == 0:
            if matrix == "laplace":
                mm_m, rp, ci, v = entry.laplace5(args.mm_grid)
                what = "5-pt Laplacian %dx%d grid" % (args.mm_grid, args.mm_grid)
            else:
                # configs[3] to the letter: "1M x 1M CSR x dense B with 256 columns, blocked-ELL MFMA tiles, B column-sharded":
                # the block-dense stand-in (16 unknowns per node, 7-point node stencil; 40^3 nodes = 1,024,000 rows)
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import standins
                e = args.bell_nodes
                mm_m, rp, ci, v = standins.block_dense(e, e, e)
                what = "block-dense stand-in, %d^3 nodes x 16 unknowns (blocked-ELL copy, MFMA kernel)" % e
            csr_mm = (mm_m, mm_m, rp, ci, v)
        res, sh, B, C = sharded.bench_sharded_csrmm(pkg, torch, D, device, rank, world, csr_mm, args.mm_cols,
                                                    layout=layout, reps=20 if matrix == "laplace" else 10, warm=3, full_product=True,
                                                    allgather=world > 1, peak_gbs=HBM_PEAK_GBS)
        what = sharded.broadcast_text(D, torch, device, rank, what if rank == 0 else None)
        res["bell_width"] = int(sh.A.spmv_info().mm_bell_width)
        res["workload"] = ("aoclsparse_dcsrmm, A = %s, B %d x %d fp64 %s, beta=0 (C read and multiplied "
                           "by zero as in the reference: the default), columns "
                           "sharded over %d rank(s) by the reference's thread-split rule (csrmm_kt.cpp:68-82); A's analysed device "
                           "arrays broadcast from rank 0, no data-path collective"
                           % (what, sh.m, args.mm_cols, res["layout"], world))
        if rank == 0:
            import oracle
            # parity of rank 0's slab: first 4 columns against the oracle's column-major reference kernel
            ns = min(4, sh.nloc)
            cols = C.reshape(sh.nloc, sh.m)[:ns] if sh.layout == "col" else C.reshape(sh.m, sh.nloc)[:, :ns].t().contiguous()
            bcol = B.reshape(sh.nloc, sh.m)[:ns] if sh.layout == "col" else B.reshape(sh.m, sh.nloc)[:, :ns].t().contiguous()
            _, Cr = oracle.dcsrmm("col", 1.0, 0, csr_mm[4], csr_mm[3], csr_mm[2], sh.m, bcol.cpu().numpy().reshape(-1),
                                  ns, sh.m, 0.0, np.zeros(ns * sh.m), sh.m)
            res["parity"] = {"vs": "oracle csrmm_col_major_ref, %d columns of rank 0's slab" % ns,
                             "bit_exact": bool(np.array_equal(cols.cpu().numpy().reshape(-1), Cr))}
        del sh, B, C
        return res

    for lay in (("col", "row") if args.mm_layout == "both" else (args.mm_layout,)):
        if "csrmm_sharded" in legs:
            legs.add("csrmm_sharded_" + lay)
            run_leg("csrmm_sharded_" + lay, lambda lay=lay: leg_csrmm_sharded(lay), collective=True)
    if "csrmm_sharded" in legs and args.bell_nodes > 0:
        legs.add("csrmm_sharded_bell")
        run_leg("csrmm_sharded_bell", lambda: leg_csrmm_sharded("col", "block-dense"), collective=True)

    # ---- SURVEY 8e "next": the iteration x <- A x with A split by rows and one all-gather of the slices per iteration ----
    def leg_spmv_row_sharded():
        import math
        # world > 1: every rank builds its own rows of a grid sized so that its share stays the headline's 1.34 GB whatever N is
        # (no rank ever holds all of A); one rank: a small grid through the broadcast-and-slice path (control flow only)
        sg = args.shard_grid or (int(4096 * math.sqrt(world)) if world > 1 else 2048)
        own_rows = world > 1 and (not args.shard_grid or args.shard_own_rows)
        csr_s = None
        if not own_rows and rank == 0:
            ms_, rp_, ci_, v_ = entry.laplace5(sg)
            csr_s = (ms_, ms_, rp_, ci_, v_)
        res, sh, y_first, x0 = sharded.bench_sharded_spmv(pkg, torch, D, device, rank, world, csr_s, iters=20, warm=3,
                                                          peak_gbs=HBM_PEAK_GBS,
                                                          build_rows=(lambda r0, r1: entry.laplace5_rows(sg, r0, r1)) if own_rows else None,
                                                          m=sg * sg)
        res["workload"] = ("aoclsparse_dmv on row slices of the 5-pt Laplacian %dx%d grid, %d rank(s)%s; every iteration ends with an "
                           "all-gather of the y slices (%s)" % (sg, sg, world, ", each rank builds its own rows" if own_rows else "",
                                                               "RCCL" if args.backend == "nccl" and world > 1 else
                                                               "gloo, CPU tensors" if world > 1 else "one rank: none"))
        if rank == 0:
            import oracle
            # rank 0's rows: a row's chain does not depend on the other rows, so the oracle on the slice IS the unsharded product
            ml, nl, rpl, cil, vl = sh.local
            so, yr = oracle.dcsrmv(-1, 0, 1.0, ml, len(vl), vl, cil, rpl, x0.cpu().numpy(), 0.0, np.zeros(ml),
                                   nthreads=oracle.max_threads())
            res["parity"] = {"vs": "oracle ref_csrmv_gn order, rank 0's rows of the UNSHARDED product",
                             "bit_exact": bool(np.array_equal(y_first.cpu().numpy(), yr))}
            unpin()
        return res

    run_leg("spmv_row_sharded", leg_spmv_row_sharded, collective=True)

    # ---- SURVEY 8e "sp2m: later": C = A * A with A's rows split over the ranks, A itself (as B) on every rank ----
    def leg_sp2m_row_sharded():
        sg = args.sp2m_grid
        ms_, rp_, ci_, v_ = entry.laplace5(sg)  # B (every rank holds it) -- and the source of the parity check
        res, sh, e = sharded.bench_sharded_sp2m(pkg, torch, D, device, rank, world, sg * sg,
                                                lambda r0, r1: entry.laplace5_rows(sg, r0, r1), (ms_, ms_, rp_, ci_, v_))
        res["workload"] = "aoclsparse_sp2m(A_r, A): the rank's rows of the 5-pt Laplacian %dx%d grid times the whole matrix" % (sg, sg)
        if rank == 0:
            import oracle
            ml, k, rpl, cil, vl = sh.local
            so, pc, ic, vc = oracle.dcsr2m(ml, ms_, int(rpl[0]), rpl, cil, vl, 0, rp_, ci_, v_)
            res["parity"] = {"vs": "oracle two-stage Gustavson on rank 0's rows", "bit_exact": bool(
                so == 0 and np.array_equal(e["row_ptr"], pc) and np.array_equal(e["col_ind"], ic) and np.array_equal(e["val"], vc))}
        return res

    run_leg("sp2m_row_sharded", leg_sp2m_row_sharded, collective=True)

    if rank == 0:
        import oracle

        # ---- parity of the timed configuration against the oracle (checker, not timed) ----
        yd = y.cpu().numpy()
        so, yref = oracle.dcsrmv(-1, 0, 1.0, m, nnz, val, col_ind, row_ptr, xh, 0.0, np.zeros(m),
                                 nthreads=oracle.max_threads())
        out["parity"] = {"vs": "oracle ref_csrmv_gn order", "bit_exact": bool(np.array_equal(yd, yref)),
                         "max_abs_diff": float(np.max(np.abs(yd - yref)))}
        unpin()

    # ---- literal configs[1]: L100 (10k x 10k), launch-latency bound ----
    def leg_l100():
        import oracle
        m1, rp1, ci1, v1 = entry.laplace5(100)
        A1 = pkg.Matrix(0, m1, m1, rp1, ci1, v1)
        assert L.aoclsparse_set_mv_hint(A1.h, pkg.OP_NONE, descr.h, 1000) == 0
        assert L.aoclsparse_optimize(A1.h) == 0
        x1 = torch.from_numpy(np.sin(0.01 * np.arange(m1))).to(device)
        y1 = torch.zeros(m1, dtype=torch.float64, device=device)
        lp = timed_laps(pkg, lambda: pkg.dmv(pkg.OP_NONE, 1.0, A1, descr, x1, 0.0, y1), 2000, 50)
        us = float(np.mean(lp)) * 1e3
        b1 = spmv_bytes(m1, m1, len(v1))
        so, yr1 = oracle.dcsrmv(-1, 0, 1.0, m1, len(v1), v1, ci1, rp1, x1.cpu().numpy(), 0.0, np.zeros(m1))
        exact = bool(np.array_equal(y1.cpu().numpy(), yr1))
        out = {"workload": "BASELINE configs[1] literal: aoclsparse_dmv, 10k x 10k 5-pt Laplacian, nnz=%d" % len(v1),
               "us_per_call": round(us, 3), "stats_ms": quartiles(lp), "gflops": round(2.0 * len(v1) / us / 1e3, 3),
               "roofline": roofline(b1, us * 1e-3), "bit_exact": exact,
               "note": "795 KB problem: bound by launch latency, not HBM"}
        inf1 = A1.spmv_info()
        out["roofline_latency"] = latency_floor(max(inf1.row_blocks, 1), b1, len(v1), us)
        # the same call back to back without an event per call, and 100 of them captured once into a HIP graph on the stream
        # handed to aoclsparse_mi355_set_stream and replayed (device-pointer calls only enqueue kernels: INTEGRATION.md)
        try:
            import ctypes
            torch.cuda.synchronize()
            pkg.timer_start()
            for _ in range(2000):
                pkg.dmv(pkg.OP_NONE, 1.0, A1, descr, x1, 0.0, y1)
            out["us_per_call_back_to_back"] = round(pkg.timer_stop() / 2000 * 1e3, 3)
            # the same with the ctypes arguments converted ONCE (what is left of Python is the foreign-function call itself)
            fn, a1, b1 = L.aoclsparse_dmv, ctypes.c_double(1.0), ctypes.c_double(0.0)
            pa, pb = ctypes.byref(a1), ctypes.byref(b1)
            px, py = ctypes.c_void_p(x1.data_ptr()), ctypes.c_void_p(y1.data_ptr())
            ah, dh, opn = A1.h, descr.h, pkg.OP_NONE
            torch.cuda.synchronize()
            pkg.timer_start()
            for _ in range(2000):
                fn(opn, pa, ah, dh, px, pb, py)
            out["us_per_call_back_to_back_prebound_arguments"] = round(pkg.timer_stop() / 2000 * 1e3, 3)
            sg = torch.cuda.Stream()
            assert L.aoclsparse_mi355_set_stream(ctypes.c_void_p(sg.cuda_stream)) == 0
            try:
                with torch.cuda.stream(sg):
                    pkg.dmv(pkg.OP_NONE, 1.0, A1, descr, x1, 0.0, y1)
                    sg.synchronize()
                    gr = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gr, stream=sg):
                        for _ in range(100):
                            pkg.dmv(pkg.OP_NONE, 1.0, A1, descr, x1, 0.0, y1)
                    gr.replay()
                    sg.synchronize()
                    t0g = time.perf_counter()
                    for _ in range(20):
                        gr.replay()
                    sg.synchronize()
                    out["us_per_call_in_a_hip_graph_of_100"] = round((time.perf_counter() - t0g) / 2000 * 1e6, 3)
                    out["graph_bit_exact"] = bool(np.array_equal(y1.cpu().numpy(), yr1))
            finally:
                assert L.aoclsparse_mi355_set_stream(None) == 0
        except Exception as e:  # noqa: BLE001 -- the graph figure is an extra
            out["hip_graph_error"] = repr(e)[:200]
        # The figures above are timed from Python (ctypes: ~1-2 us of interpreter per call).  What a C caller pays, next to a raw
        # launch of an empty kernel with the same 13 arguments: tools/l100_probe.hip, run as a child process with a timeout.
        try:
            import subprocess
            probe = os.path.join(ROOT, "tools", "bin", "l100_probe")
            if os.path.exists(probe):
                r = subprocess.run([probe, "5000"], capture_output=True, text=True, timeout=120)
                line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
                out["c_caller"] = json.loads(line[-1]) if r.returncode == 0 and line else {"error": (r.stdout + r.stderr)[-200:]}
            else:
                out["c_caller"] = {"skipped": "tools/bin/l100_probe not built (python -c 'import __graft_entry__ as g; g.build()')"}
        except Exception as e:  # noqa: BLE001 -- an extra
            out["c_caller"] = {"error": repr(e)[:200]}
        return out

    run_leg("l100", leg_l100)

    # ---- raw aoclsparse_dcsrmv on device-resident CSR arrays: the CSR-Adaptive kernel (configs[1] wording) ----
    def leg_csr_adaptive():
        d_rp, d_ci, d_v = (torch.from_numpy(a).to(device) for a in (row_ptr, col_ind, val))
        y2 = torch.zeros(m, dtype=torch.float64, device=device)

        def call():
            s = pkg.dcsrmv(pkg.OP_NONE, 1.0, m, m, nnz, d_v, d_ci, d_rp, descr, x, 0.0, y2)
            assert s == 0, pkg.STATUS[s]
        # cold products, as the headline (the row-block kernel alternates its block order too); the back-to-back mean beside it
        flush = torch.ones(1 << 28, dtype=torch.float32, device=device)
        lp = timed_cold(pkg, call, min(args.steps, 30), flush)
        del flush
        ms = float(np.mean(lp))
        ms_b2b = float(np.mean(timed_laps(pkg, call, min(args.steps, 30), 3)))
        tr_ca = None
        try:  # the committed PMC measurement of this kernel on this workload
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                pj = json.load(f)
            if pj.get("grid") == g:
                tr_ca = (pj.get("csr_adaptive_kernel") or pj.get("previous_kernel") or {}).get("traffic_bytes_per_launch")
        except (OSError, ValueError):
            pass
        return {"workload": "aoclsparse_dcsrmv (no handle), CSR arrays / x / y device-resident, same %dx%d-grid Laplacian" % (g, g),
                "kernel": "csr-adaptive (row blocks staged in LDS)", "ms": round(ms, 6), "stats_ms": quartiles(lp),
                "timing": "cache flushed before every product", "ms_back_to_back": round(ms_b2b, 6),
                "gflops": round(flops / ms / 1e6, 2),
                "roofline": roofline(abytes, ms, tr_ca, traffic_source="profiles/pmc_traffic.json" if tr_ca else None),
                "bit_exact_vs_headline_y": bool(torch.equal(y2, y))}

    run_leg("dcsrmv_csr_adaptive", leg_csr_adaptive)

    # ---- the same headline matrix in fp32 and through the literal drop-in call with HOST vectors ----
    def leg_headline_twins():
        pmc = {}
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                pmc = json.load(f)
        except (OSError, ValueError):
            pass
        res = {}
        # aoclsparse_smv: float values, the reference's 8-lane float order (csrmv_kr.hpp:734-831); bytes model of the reference
        # harness with 4-byte values (aoclsparse_gbyte.hpp:39-45)
        vf = val.astype(np.float32)
        Af = pkg.Matrix(0, m, m, row_ptr, col_ind, vf)
        assert L.aoclsparse_set_mv_hint(Af.h, pkg.OP_NONE, descr.h, 1000) == 0 and L.aoclsparse_optimize(Af.h) == 0
        xf = x.to(torch.float32)
        yf = torch.zeros(m, dtype=torch.float32, device=device)
        flush = torch.ones(1 << 28, dtype=torch.float32, device=device)
        lp = timed_cold(pkg, lambda: pkg.smv(pkg.OP_NONE, 1.0, Af, descr, xf, 0.0, yf), min(args.steps, 30), flush)  # cold, as the headline
        ms = float(np.mean(lp))
        ms_b2b = float(np.mean(timed_laps(pkg, lambda: pkg.smv(pkg.OP_NONE, 1.0, Af, descr, xf, 0.0, yf), min(args.steps, 30), 3)))
        fbytes = (m + 1 + nnz) * 4 + (2 * m + nnz) * 4
        ftr = (pmc.get("float_kernel") or {}).get("traffic_bytes_per_launch") if pmc.get("grid") == g else None
        import oracle
        nchk = min(m, 1 << 20)  # the first 2^20 rows against the oracle's float restatement (they touch only their own neighbours)
        last = int(row_ptr[nchk])
        # (the reference's only float kernel: 8 lanes + scalar tail, csrmv.hpp:317-321)
        so, yref = oracle.scsrmv("lane8", 0, 1.0, nchk, vf[:last], col_ind[:last], row_ptr[:nchk + 1], xf.cpu().numpy(), 0.0,
                                 np.zeros(nchk, np.float32))
        torch.cuda.synchronize()
        res["smv"] = {"workload": "aoclsparse_smv, same %dx%d-grid Laplacian, float values" % (g, g), "ms": round(ms, 6),
                      "stats_ms": quartiles(lp), "gflops": round(flops / ms / 1e6, 2), "timing": "cache flushed before every product",
                      "ms_back_to_back": round(ms_b2b, 6),
                      # (frac on CSR-model bytes passes 1: with float values the shared column lists are 0.36 of the model's bytes;
                      # frac_traffic is the fraction in real bytes)
                      "roofline": roofline(fbytes, ms, ftr, traffic_source="profiles/pmc_traffic.json: float_kernel" if ftr else None),
                      "bit_exact_first_2e20_rows": bool(so == 0 and np.array_equal(yf[:nchk].cpu().numpy(), yref))}
        del Af, xf, yf, flush
        # aoclsparse_dmv exactly as a reference user calls it: x and y are HOST arrays (staged per call: 134 MB each way), the call
        # returns when y is in host memory.  Wall clock per call; the PCIe-inclusive rate, never the headline value.
        xh2, yh2 = np.ascontiguousarray(xh), np.zeros(m)
        L.aoclsparse_mi355_set_pointer_mode(pkg.PTR_AUTO)
        try:
            for _ in range(2):
                assert pkg.dmv(pkg.OP_NONE, 1.0, A, descr, xh2, 0.0, yh2) == 0
            t = []
            for _ in range(5):
                t0h = time.perf_counter()
                assert pkg.dmv(pkg.OP_NONE, 1.0, A, descr, xh2, 0.0, yh2) == 0
                t.append((time.perf_counter() - t0h) * 1e3)
        finally:
            L.aoclsparse_mi355_set_pointer_mode(pkg.PTR_DEVICE)
        # the headline product COLD: 1 GB of other data is READ between any two timed products, so that nothing of the matrix is
        # left in the 256 MB Infinity Cache from the call before (back-to-back products keep part of their working set there;
        # profiles/r5/sell_placement.txt).  One product per event pair.  (A 1 GB FILL as the flush leaves dirty lines whose
        # write-back runs into the product: 0.211 instead of 0.173 ms, tools/history/exp_cold.py.)  Never the headline value.
        try:
            flush = torch.ones(1 << 28, dtype=torch.float32, device=device)
            tc = []
            for _ in range(12):
                flush.sum()
                torch.cuda.synchronize()
                pkg.timer_start()
                assert pkg.dmv(pkg.OP_NONE, 1.0, A, descr, x, 0.0, y) == 0
                tc.append(pkg.timer_stop())
            del flush
            cms = float(np.median(tc[2:]))
            res["cold_dmv"] = {"workload": "the headline aoclsparse_dmv with the Infinity Cache flushed before every product (1 GB read)",
                               "ms": round(cms, 6), "roofline": roofline(abytes, cms, traffic)}
        except Exception as e:  # noqa: BLE001 -- an extra
            res["cold_dmv"] = {"error": repr(e)[:200]}
        res["host_pointer_dmv"] = {"workload": "aoclsparse_dmv with host x / y (reference calling convention), same matrix",
                                   "ms_wall_per_call": round(float(np.median(t)), 3), "bytes_over_pcie_per_call": 16 * m,
                                   "bit_exact_vs_headline_y": bool(np.array_equal(yh2, y.cpu().numpy()))}
        return res

    run_leg("headline_twins", leg_headline_twins)

    # ---- configs[2]: the SuiteSparse mix through set_mv_hint + optimize ----
    def leg_mix():
        import oracle
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import standins
        rows = []
        pmc_irr, pmc_irr_src = {}, None
        for rnd in ("r6", "r5", "r2"):  # fabric traffic of the mix kernels (PMC cannot be read from inside the run): the latest committed pass
            try:
                with open(os.path.join(ROOT, "profiles", rnd, "irregular_traffic.json")) as f:
                    pmc_irr, pmc_irr_src = json.load(f), "profiles/%s/irregular_traffic.json" % rnd
                break
            except (OSError, ValueError):
                continue
        # the four stand-ins, then (round 3) each one OFF its ideal ordering: graphs with locality, meshes with irregular
        # valence and a windowed random node order (tools/standins.py)
        flush = torch.ones(1 << 28, dtype=torch.float32, device=device)
        names = (["circuit-like", "web-like"] + ([] if args.small else ["shell-like", "flan-like"])
                 + ["circuit-like, local", "web-like, local"]
                 + ([] if args.small else ["shell-like, unstructured", "flan-like, unstructured"]))
        for name in names:
            label, mm_, rp, ci, v = standins.load(name)
            nz = len(v)
            Am = pkg.Matrix(0, mm_, mm_, rp, ci, v)
            assert L.aoclsparse_set_mv_hint(Am.h, pkg.OP_NONE, descr.h, 1000) == 0 and L.aoclsparse_optimize(Am.h) == 0
            inf = Am.spmv_info()
            xr = np.random.default_rng(1).uniform(-1, 1, mm_)
            xd = torch.from_numpy(xr).to(device)
            ydv = torch.zeros(mm_, dtype=torch.float64, device=device)
            lp = timed_laps(pkg, lambda: pkg.dmv(pkg.OP_NONE, 1.0, Am, descr, xd, 0.0, ydv), 200, 20)
            # the laps put one event record between calls (~3 us: as much as a quarter of the 10-30 us kernels here);
            # "us" is the same 200 calls back to back between two events, the laps give the spread
            torch.cuda.synchronize()
            pkg.timer_start()
            for _ in range(200):
                pkg.dmv(pkg.OP_NONE, 1.0, Am, descr, xd, 0.0, ydv)
            ms = pkg.timer_stop() / 200
            ms_lap = float(np.mean(lp))
            # the HBM-only twin: the Infinity Cache flushed before every product (an event pair per product: ~3 us of event cost
            # inside the 10-30 us of the two small matrices)
            cold_ms = float(np.median(timed_cold(pkg, lambda: pkg.dmv(pkg.OP_NONE, 1.0, Am, descr, xd, 0.0, ydv), 12, flush)))
            so, yr = oracle.dcsrmv(-1, 0, 1.0, mm_, nz, v, ci, rp, xr, 0.0, np.zeros(mm_), nthreads=oracle.max_threads())
            got = ydv.cpu().numpy()
            lens = np.diff(rp)
            # rows the automatic mode keeps in the reference's order (CSR-Adaptive: fewer than tree_min entries; SELL-64: all)
            within = lens < inf.tree_min if inf.tree_min > 0 else np.ones(mm_, bool)
            strict = None
            if inf.tree_min > 0:  # the same product with every row in the reference's order (spmv_strict: bit-exact everywhere)
                assert L.aoclsparse_mi355_set_option(pkg.OPTION_SPMV_STRICT, 1) == 0
                try:
                    ys = torch.zeros(mm_, dtype=torch.float64, device=device)
                    for _ in range(20):
                        pkg.dmv(pkg.OP_NONE, 1.0, Am, descr, xd, 0.0, ys)
                    torch.cuda.synchronize()
                    pkg.timer_start()
                    for _ in range(200):
                        pkg.dmv(pkg.OP_NONE, 1.0, Am, descr, xd, 0.0, ys)
                    ms_strict = pkg.timer_stop() / 200
                    strict = {"us": round(ms_strict * 1e3, 3), "roofline": roofline(spmv_bytes(mm_, mm_, nz), ms_strict),
                              "bit_exact_every_row": bool(np.array_equal(ys.cpu().numpy(), oracle.dcsrmv(
                                  -1, 0, 1.0, mm_, nz, v, ci, rp, xr, 0.0, np.zeros(mm_), nthreads=oracle.max_threads())[1]))}
                finally:
                    assert L.aoclsparse_mi355_set_option(pkg.OPTION_SPMV_STRICT, 0) == 0
            scale = np.zeros(mm_)
            nzr = lens > 0
            scale[nzr] = np.add.reduceat(np.abs(v * xr[ci]), rp[:-1][nzr])
            err = np.abs(got - yr)
            bound = (2 * np.ceil(np.log2(np.maximum(lens, 2))) + 4 + lens / 256.0) * np.finfo(np.float64).eps * scale
            b = spmv_bytes(mm_, mm_, nz)
            tr = (pmc_irr.get(label) or {}).get("traffic_bytes_per_launch")
            stc, secs, _ = oracle.dcsrmv_bench(-1, 0, mm_, mm_, nz, v, ci, rp, xr,
                                               max(1, min(cpu_physical, oracle.max_threads())), 5)
            unpin()
            rows.append({"matrix": label, "m": mm_, "nnz": nz, "max_row": int(lens.max()),
                         "kernel": {1: "csr-adaptive", 2: "merge-path", 3: "sell-64", 4: "sell-64, shared column lists"}.get(inf.kernel, str(inf.kernel)),
                         "summation_order": {0: "scalar (ref_csrmv_gn)", 1: "4-lane AVX2", 2: "8-lane AVX-512"}.get(inf.order),
                         "us": round(ms * 1e3, 3), "us_timing": "200 calls back to back between two events",
                         "us_per_call_with_an_event_each": round(ms_lap * 1e3, 3), "stats_ms": quartiles(lp),
                         "gflops": round(2.0 * nz / ms / 1e6, 2),
                         "roofline": roofline(b, ms, tr, traffic_source=pmc_irr_src if tr else None,
                                              traffic_gbs=round(tr / ms / 1e6, 1) if tr else None,
                                              traffic_frac_of_peak=round(tr / ms / 1e6 / HBM_PEAK_GBS, 4) if tr else None),
                         # back to back the working set of the small matrices never leaves the L2s / the Infinity Cache: `roofline`
                         # (back-to-back time over the HBM roof) is an HBM fraction only where bound_back_to_back says "hbm";
                         # `roofline_cold` always is (every byte comes from HBM)
                         "bound_back_to_back": bound_of(b), "cold_us": round(cold_ms * 1e3, 3),
                         "roofline_cold": roofline(b, cold_ms, tr, traffic_source=pmc_irr_src if tr else None),
                         "bit_exact_rows_below_tree_min": bool(np.array_equal(got[within], yr[within])), "tree_min": int(inf.tree_min),
                         "rows_outside_bit_exact_regime": int((~within).sum()), "strict_mode": strict,
                         "long_rows_within_bound": bool(np.all(err <= bound + 1e-300)),
                         "max_abs_diff": float(err.max()),
                         "cpu_all_cores_gflops": round(2.0 * nz / float(np.median(secs)) / 1e9, 2)})
            if b < (200 << 20):  # the matrix stays in the L2s / Infinity Cache between calls: state the latency roof beside HBM
                rows[-1]["roofline_latency"] = latency_floor(max(inf.row_blocks, 1), b, nz, ms * 1e3,
                                                             lanes=128 if inf.tile == 512 else 256)
            del Am, xd, ydv
        del flush
        # the third kernel aoclsparse_optimize can choose: merge-path, for matrices whose longest row spans tens of LDS
        # tiles (none of the four above does): a tridiagonal matrix with four rows of ~170 k entries
        n3 = 300000
        rng = np.random.default_rng(3)
        longs = set(int(t) for t in rng.choice(n3, size=4, replace=False))
        parts, rp3 = [], np.zeros(n3 + 1, np.int64)
        tri = np.stack([np.arange(n3) - 1, np.arange(n3), np.arange(n3) + 1], axis=1)
        for i in range(n3):
            c = np.unique(np.concatenate([rng.integers(0, n3, size=250000), [i]])) if i in longs else tri[i][(tri[i] >= 0) & (tri[i] < n3)]
            parts.append(c)
            rp3[i + 1] = rp3[i] + len(c)
        ci3 = np.concatenate(parts).astype(np.int32)
        rp3 = rp3.astype(np.int32)
        v3 = rng.uniform(-1, 1, len(ci3))
        A3 = pkg.Matrix(0, n3, n3, rp3, ci3, v3)
        assert L.aoclsparse_set_mv_hint(A3.h, pkg.OP_NONE, descr.h, 1000) == 0 and L.aoclsparse_optimize(A3.h) == 0
        x3 = rng.uniform(-1, 1, n3)
        x3d, y3d = torch.from_numpy(x3).to(device), torch.zeros(n3, dtype=torch.float64, device=device)
        lp = timed_laps(pkg, lambda: pkg.dmv(pkg.OP_NONE, 1.0, A3, descr, x3d, 0.0, y3d), 50, 5)
        so, y3r = oracle.dcsrmv(0, 0, 1.0, n3, len(v3), v3, ci3, rp3, x3, 0.0, np.zeros(n3))
        lens3 = np.diff(rp3)
        sc3 = np.add.reduceat(np.abs(v3 * x3[ci3]), rp3[:-1])
        e3 = np.abs(y3d.cpu().numpy() - y3r)
        inf3 = A3.spmv_info()
        demo = {"matrix": "tridiagonal + 4 rows of ~170 k entries (m=%d, nnz=%d)" % (n3, len(v3)),
                "kernel": {1: "csr-adaptive", 2: "merge-path", 3: "sell-64"}.get(inf3.kernel), "us": round(float(np.mean(lp)) * 1e3, 2),
                "roofline": roofline(spmv_bytes(n3, n3, len(v3)), float(np.mean(lp))),
                "rows_not_bit_exact": int(np.sum(y3d.cpu().numpy() != y3r)), "merge_tiles": int((n3 + len(v3)) // 1024 + 1),
                "note": "a row is bit-exact unless a 1,024-item tile boundary cuts it (at most one row per tile)",
                "all_rows_within_bound": bool(np.all(e3 <= (lens3 + lens3 / 1024.0 + 8) * np.finfo(np.float64).eps * sc3 + 1e-300))}
        unpin()
        return {"workload": "BASELINE configs[2]: aoclsparse_dmv after aoclsparse_set_mv_hint + aoclsparse_optimize "
                            "(format / kernel chosen by optimize from the row-length statistics: SELL-64 when padding "
                            "<= 1.35x, merge-path when the longest row spans >= 32 LDS tiles, else CSR-Adaptive)",
                "matrices": rows, "merge_path_selection": demo}

    run_leg("mix", leg_mix)

    # ---- configs[3] on one GPU: both layouts, 256 columns and a 32-column slab ----
    def leg_csrmm():
        import oracle
        mm_m, rp, ci, v = entry.laplace5(args.mm_grid)
        nz = len(v)
        res = {"workload": "aoclsparse_dcsrmm, A = 5-pt Laplacian %dx%d grid (m=%d, nnz=%d), beta=0, B U(-1,1) "
                           "column j seeded 777+j" % (args.mm_grid, args.mm_grid, mm_m, nz), "cases": []}
        flush = torch.ones(1 << 28, dtype=torch.float32, device=device)
        for layout in ("row", "col"):
            sh = sharded.ShardedCsrmm(pkg, torch, None, device, 0, 1, (mm_m, mm_m, rp, ci, v), args.mm_cols, layout)
            B = sh.make_B()
            C = torch.zeros(args.mm_cols * mm_m, dtype=torch.float64, device=device)
            j0, j1 = column_shard(args.mm_cols, 8, 0)
            for ncols, tag in ((args.mm_cols, "all %d columns" % args.mm_cols), (j1 - j0, "one slab of an 8-rank run")):
                Bs = B if ncols == args.mm_cols else sh.make_B(j0=j0, j1=j1)
                Cs = C if ncols == args.mm_cols else torch.zeros(ncols * mm_m, dtype=torch.float64, device=device)
                # beta = 0 twice: the default reads C and multiplies it by zero, as every reference kernel does (NaN / Inf in C
                # propagate: csrmm.hpp:83,129) -- its algorithmic bytes therefore include the read of C; the opt-in overwrite
                # mode (aoclsparse_mi355_set_csrmm_beta0_overwrite) does not read C
                ns = min(4, ncols)
                bb = Bs.reshape(ncols, mm_m)[:ns] if layout == "col" else Bs.reshape(mm_m, ncols)[:, :ns].t().contiguous()
                bb_host = bb.cpu().numpy().reshape(-1)
                for beta, overwrite in ((0.0, False), (0.0, True), (-2.0, False)):
                    assert L.aoclsparse_mi355_set_csrmm_beta0_overwrite(1 if overwrite else 0) == 0
                    lp = timed_laps(pkg, lambda: sh.run(Bs, Cs, beta=beta, nloc=ncols), 20, 3)
                    cold_ms = float(np.median(timed_cold(pkg, lambda: sh.run(Bs, Cs, beta=beta, nloc=ncols), 8, flush)))
                    # parity of THIS mode: C preset to 0.25, one product, 4 columns against the oracle's column-major reference
                    # kernel with the same beta and the same C (overwrite mode: identical results for finite C)
                    Cs.fill_(0.25)
                    assert sh.run(Bs, Cs, beta=beta, nloc=ncols) == 0
                    torch.cuda.synchronize()
                    assert L.aoclsparse_mi355_set_csrmm_beta0_overwrite(0) == 0
                    cc = Cs.reshape(ncols, mm_m)[:ns] if layout == "col" else Cs.reshape(mm_m, ncols)[:, :ns].t().contiguous()
                    _, Cr = oracle.dcsrmm("col", 1.0, 0, v, ci, rp, mm_m, bb_host, ns, mm_m, beta, np.full(ns * mm_m, 0.25), mm_m)
                    ms = float(np.mean(lp))
                    reads_c = beta != 0.0 or not overwrite
                    b = csrmm_bytes(mm_m, mm_m, nz, ncols, reads_c)
                    res["cases"].append({"layout": "row-major" if layout == "row" else "column-major", "ncols": ncols,
                                         "what": tag, "beta": beta,
                                         "c_is_read": reads_c, "mode": ("default: C read (reference arithmetic)" if beta == 0.0 and not overwrite
                                                                        else "opt-in: C overwritten" if beta == 0.0 else "beta != 0"),
                                         "ms": round(ms, 5), "stats_ms": quartiles(lp),
                                         "gflops": round(2.0 * nz * ncols / ms / 1e6, 1), "roofline": roofline(b, ms),
                                         # `roofline` is the back-to-back loop (a slab's 0.6-0.8 GB mostly stays in no cache, the
                                         # alternating block order keeps the end of C / B in the Infinity Cache); `roofline_cold`:
                                         # cache flushed before every product -- the HBM fraction
                                         "bound_back_to_back": bound_of(b), "cold_ms": round(cold_ms, 5), "roofline_cold": roofline(b, cold_ms),
                                         "roofline_survey_model": roofline(csrmm_bytes(mm_m, mm_m, nz, ncols, beta != 0.0), ms),
                                         "bit_exact_4_columns": bool(np.array_equal(cc.cpu().numpy().reshape(-1), Cr))})
                # a pinned kid (row-major): the KT kernels' arithmetic (csrmm_row_kt), on the tuned kernels when the column count
                # is a multiple of the vector width; first 8 columns checked against the KT restatement (4 / 8 lanes)
                if layout == "row":
                    ld = ncols
                    for kid, lanes in ((1, 4), (3, 8)):
                        call = lambda: L.aoclsparse_dcsrmm_kid(pkg.OP_NONE, 1.0, sh.A.h, sh.descr.h, pkg.ORDER_ROW, pkg._ptr(Bs), ncols, ld,
                                                               0.0, pkg._ptr(Cs), ld, kid)
                        Cs.zero_()
                        lp = timed_laps(pkg, call, 10, 2)
                        ms = float(np.mean(lp))
                        b = csrmm_bytes(mm_m, mm_m, nz, ncols, True)
                        ent = {"layout": "row-major", "ncols": ncols, "what": tag, "beta": 0.0, "c_is_read": True,
                               "mode": "kid %d: csrmm_row_kt order, %d lanes (%s)" % (kid, lanes, "AVX2 host" if kid == 1 else "AVX-512 host"),
                               "ms": round(ms, 5), "stats_ms": quartiles(lp), "gflops": round(2.0 * nz * ncols / ms / 1e6, 1),
                               "roofline": roofline(b, ms)}
                        if ncols % 8 == 0:
                            # the kernel's arithmetic does not depend on which columns are present: compare 8 columns of a
                            # separate 8-column product (same kernel family, kid as above) with the oracle
                            B8 = Bs.reshape(mm_m, ncols)[:, :8].contiguous()
                            C8 = torch.zeros(mm_m * 8, dtype=torch.float64, device=device)
                            assert L.aoclsparse_dcsrmm_kid(pkg.OP_NONE, 1.0, sh.A.h, sh.descr.h, pkg.ORDER_ROW, pkg._ptr(B8), 8, 8, 0.0,
                                                           pkg._ptr(C8), 8, kid) == 0
                            torch.cuda.synchronize()
                            _, Ck = oracle.dcsrmm_kt("row", lanes, 1.0, 0, v, ci, rp, mm_m, B8.cpu().numpy().reshape(-1), 8, 8, 0.0,
                                                     np.zeros(8 * mm_m), 8)
                            full8 = Cs.reshape(mm_m, ncols)[:, :8].contiguous().cpu().numpy().reshape(-1)
                            ent["bit_exact_8_columns_vs_kt_oracle"] = bool(np.array_equal(C8.cpu().numpy(), Ck)
                                                                           and np.array_equal(full8, Ck))
                        res["cases"].append(ent)
            del sh, B, C
        del flush
        try:
            res["blocked"] = leg_csrmm_blocked()
        except Exception as e:  # noqa: BLE001 -- an extra of the csrmm leg
            res["blocked"] = {"error": "%s: %s" % (type(e).__name__, e)}
        return res

    # ---- north_star's "blocked-ELL variant that feeds MFMA where tiles are dense": a block-dense stand-in (16 unknowns per node,
    # tile fill 1.0 and 0.75), row-major, the blocked-ELL copy on v_mfma_f64_16x16x4_f64 (what aoclsparse_optimize builds for the
    # mm hint) against the CSR kernels on the SAME matrix (a handle under aoclsparse_memory_usage_minimal gets no second copy)
    def leg_csrmm_blocked():
        import oracle
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import standins
        out = {"workload": "aoclsparse_dcsrmm row-major, %d columns, beta=0 (C read), block-dense stand-ins: 16 unknowns per node, "
                           "7-point node stencil" % args.mm_cols, "cases": []}
        ncols = args.mm_cols
        for name in (("block-dense",) if args.small else ("block-dense", "block-dense, 75 % fill")):
            if args.small:
                label, (mb, rp, ci, v) = name + " (stand-in, 8x8x8 nodes)", standins.block_dense(8, 8, 8, keep=1.0)
            else:
                label, mb, rp, ci, v = standins.load(name)
            nz = len(v)
            Bd = sharded.make_B_slab(torch, device, mb, 0, ncols, "row")
            Cd = torch.zeros(mb * ncols, dtype=torch.float64, device=device)
            bcol4 = Bd.reshape(mb, ncols)[:, :4].t().contiguous().cpu().numpy().reshape(-1)
            _, Cr = oracle.dcsrmm("col", 1.0, 0, v, ci, rp, mb, bcol4, 4, mb, 0.0, np.zeros(4 * mb), mb)
            ent = {"matrix": label, "m": mb, "nnz": nz, "ncols": ncols}
            for kind in ("mfma", "csr"):
                Ab = pkg.Matrix(0, mb, mb, rp, ci, v)
                if kind == "csr":
                    assert L.aoclsparse_set_memory_hint(Ab.h, 0) == 0  # aoclsparse_memory_usage_minimal: no second copy
                assert L.aoclsparse_set_mm_hint(Ab.h, pkg.OP_NONE, descr.h, 100) == 0 and L.aoclsparse_optimize(Ab.h) == 0
                call = lambda: pkg.dcsrmm(pkg.OP_NONE, 1.0, Ab, descr, pkg.ORDER_ROW, Bd, ncols, ncols, 0.0, Cd, ncols)
                assert call() == 0
                lp = timed_laps(pkg, call, 10, 2)
                torch.cuda.synchronize()
                inf = Ab.spmv_info()
                ms = float(np.mean(lp))
                got = Cd.reshape(mb, ncols)[:, :4].t().contiguous().cpu().numpy().reshape(-1)
                tf = 2.0 * nz * ncols / ms / 1e9
                ent[kind] = {"ms": round(ms, 5), "tflops": round(tf, 2),
                             # the bound of the MFMA kernel is the matrix pipe, not HBM: dense fp64 MFMA peak of the part (spec) and
                             # what back-to-back v_mfma_f64_16x16x4_f64 sustains on it (tools/mfma_f64_peak.hip,
                             # profiles/r4/mfma_f64_peak.jsonl: 47.7 TFLOP/s with >= 2 wavefronts per SIMD)
                             "roofline_mfma": {"bound": "mfma", "achieved": round(tf, 2), "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                               "frac": round(tf / FP64_MFMA_PEAK_TFLOPS, 4),
                                               "sustained_back_to_back": FP64_MFMA_SUSTAINED_TFLOPS,
                                               "frac_of_sustained": round(tf / FP64_MFMA_SUSTAINED_TFLOPS, 4)} if kind == "mfma" else None,
                             "bell_width": int(inf.mm_bell_width), "tile_fill": inf.mm_bell_fill_permille / 1000.0,
                             "roofline": roofline(csrmm_bytes(mb, mb, nz, ncols, True), ms),
                             "bit_exact_4_columns": bool(np.array_equal(got, Cr))}
                if kind == "mfma":
                    # which XCD works through which block rows (round 6: csrmm_api.cpp choose_bell_order) and the opt-in mode that does
                    # not read C, where the kernel is bound by the matrix pipe rather than by the fabric
                    ent[kind]["block_row_order"] = {
                        "xcd_chunk": int(inf.mm_bell_xcd_chunk), "lattice": [int(inf.mm_bell_lattice_line), int(inf.mm_bell_lattice_lines)],
                        "region": [int(inf.mm_bell_region_a), int(inf.mm_bell_region_b)],
                        "model_fetches_per_b_block_row": inf.mm_bell_model_fetches_permille / 1000.0,
                        "model_fetches_launch_order": inf.mm_bell_model_fetches_launch_order_permille / 1000.0}
                    assert L.aoclsparse_mi355_set_csrmm_beta0_overwrite(1) == 0
                    try:
                        assert call() == 0
                        ms_ow = float(np.mean(timed_laps(pkg, call, 10, 2)))
                    finally:
                        assert L.aoclsparse_mi355_set_csrmm_beta0_overwrite(0) == 0
                    ent[kind]["overwrite"] = {"ms": round(ms_ow, 5), "tflops": round(2.0 * nz * ncols / ms_ow / 1e9, 2),
                                              "frac_of_sustained": round(2.0 * nz * ncols / ms_ow / 1e9 / FP64_MFMA_SUSTAINED_TFLOPS, 4)}
                del Ab
            # the same matrix with column-major operands (the layout the column shards are contiguous in): all columns and the
            # 32-column slab of an 8-rank run -> the projected efficiency of configs[3]'s "blocked-ELL MFMA tiles, B column-sharded"
            Ab = pkg.Matrix(0, mb, mb, rp, ci, v)
            assert L.aoclsparse_set_mm_hint(Ab.h, pkg.OP_NONE, descr.h, 100) == 0 and L.aoclsparse_optimize(Ab.h) == 0
            j0, j1 = column_shard(ncols, 8, 0)
            for nc, tag in ((ncols, "mfma_col"), (j1 - j0, "mfma_col_slab")):
                Bc = sharded.make_B_slab(torch, device, mb, 0, nc, "col")
                Cc = torch.zeros(mb * nc, dtype=torch.float64, device=device)
                call = lambda: pkg.dcsrmm(pkg.OP_NONE, 1.0, Ab, descr, pkg.ORDER_COLUMN, Bc, nc, mb, 0.0, Cc, mb)
                assert call() == 0
                lp = timed_laps(pkg, call, 10, 2)
                torch.cuda.synchronize()
                ms = float(np.mean(lp))
                got = Cc.reshape(nc, mb)[:4].cpu().numpy().reshape(-1)
                ent[tag] = {"ms": round(ms, 5), "ncols": nc, "tflops": round(2.0 * nz * nc / ms / 1e9, 2),
                            "bit_exact_4_columns": bool(np.array_equal(got, Cr))}
                del Bc, Cc
            ent["mfma_col_eff8"] = _eff8(ent["mfma_col"]["ms"], ent["mfma_col_slab"]["ms"])
            del Ab
            ent["mfma_selected_by_optimize"] = ent["mfma"]["bell_width"] > 0 and ent["csr"]["bell_width"] == 0
            ent["speedup"] = round(ent["csr"]["ms"] / ent["mfma"]["ms"], 3)
            out["cases"].append(ent)
            del Bd, Cd
        out["mfma_ms"] = out["cases"][0]["mfma"]["ms"]
        out["best_other_ms"] = out["cases"][0]["csr"]["ms"]
        out["mfma_col_ms"] = out["cases"][0]["mfma_col"]["ms"]
        out["mfma_col_eff8"] = out["cases"][0]["mfma_col_eff8"]
        out["parity_ok"] = all(c["mfma"]["bit_exact_4_columns"] and c["csr"]["bit_exact_4_columns"] and c["mfma_col"]["bit_exact_4_columns"]
                               and c["mfma_col_slab"]["bit_exact_4_columns"] for c in out["cases"])
        out["fp64_mfma_peak_tflops"] = FP64_MFMA_PEAK_TFLOPS
        out["fp64_mfma_sustained_tflops"] = FP64_MFMA_SUSTAINED_TFLOPS
        return out

    run_leg("csrmm", leg_csrmm)

    # ---- configs[4]: unit-lower ILU(0) factor of the shell-like matrix ----
    def leg_trsv():
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import standins
        if args.small:
            return trsv_system("ILU(0) of 5-pt Laplacian grid 300^2", *entry.laplace5(300))
        label, mt, rp, ci, v = standins.load("shell-like")
        res = trsv_system("ILU(0) factor of %s" % label, mt, rp, ci, v)
        # round 3: the same mesh OFF its ideal ordering (10 % of the couplings dropped, nodes renumbered inside windows): how
        # the supernodal schedule (chains of dof rows) and the level structure hold up
        label2, mt, rp, ci, v = standins.load("shell-like, unstructured")
        res["unstructured_variant"] = trsv_system("ILU(0) factor of %s" % label2, mt, rp, ci, v, kids=(-1,))
        return res

    def trsv_system(title, mt, rp, ci, v, kids=(-1, 3, 1)):
        import oracle
        t = time.perf_counter()
        stf, lu, dg = oracle.dilu0(mt, 0, rp, ci, v)  # input preparation (the reference factorises on the CPU too)
        t_ilu = time.perf_counter() - t
        assert stf == 0
        At = pkg.Matrix(0, mt, mt, rp, ci, lu)
        dl = pkg.Descr(mtype=pkg.TYPE_TRIANGULAR, fill=pkg.FILL_LOWER, diag=pkg.DIAG_UNIT)
        t = time.perf_counter()
        assert L.aoclsparse_set_sv_hint(At.h, pkg.OP_NONE, dl.h, 100) == 0 and L.aoclsparse_optimize(At.h) == 0
        t_opt = time.perf_counter() - t
        lv = At.trsv_levels(pkg.FILL_LOWER)
        o = oracle.dcsr_optimize(mt, mt, len(lu), 0, rp, ci, lu)
        nnz_l = int(np.sum(o["idiag"] - rp[:-1]))
        # b = L * 1: the exact solution is the vector of ones
        ones = np.ones(mt)
        _, bh = oracle.dcsrmv_special("tri", 0, 1.0, mt, mt, 1, 0, lu, ci, rp, o["idiag"], o["iurow"], ones, 0.0, np.zeros(mt))
        t = time.perf_counter()
        _, xr = oracle.dtrsv("l", 1.0, mt, 0, lu, ci, rp, o["idiag"], bh, True)
        t_cpu = time.perf_counter() - t
        bdev = torch.from_numpy(bh).to(device)
        xdev = torch.zeros(mt, dtype=torch.float64, device=device)
        ab = trsv_bytes(mt, nnz_l)
        res = {"workload": "BASELINE configs[4]: aoclsparse_dtrsv, %s (m=%d, %d strict-lower entries, %d dependency "
                           "levels), descr {triangular, lower, unit}, alpha=1, b = L*1" % (title, mt, nnz_l, lv),
               "schedules": [], "cpu_serial_ms": round(t_cpu * 1e3, 3), "analysis_s": round(t_opt, 2),
               "ilu0_input_preparation_s": round(t_ilu, 2)}
        ti = At.trsv_info(pkg.FILL_LOWER)
        # what the plan holds and which schedule its model picked (5 = two levels: chunks of consecutive blocks, hand-offs through
        # LDS inside a chunk; 4 = a lane per block of chained rows, every hand-off through L2 / HBM)
        res["plan"] = {"row_levels": ti.levels, "blocks": ti.blocks, "block_levels": ti.block_levels, "chunks": ti.chunks, "steps": ti.steps,
                       "lds_words_largest_chunk": ti.lds_slots, "model_two_level_us": ti.model_chunk_us,
                       "model_lane_per_block_us": ti.model_block_us, "automatic_schedule": ti.schedule}
        # the kid selects the ARITHMETIC (as in the reference, trsv.cpp:321-353): auto / 0 = ref_trsv_l's chain on the fastest
        # schedule; 3 = the order of the 512-bit KT kernel an AVX-512 host dispatches, 1 = the 256-bit one (both served by the
        # lane-per-position sync-free kernel), each checked bit for bit against the oracle's restatement of THAT kernel
        _, xk8 = oracle.trsv_kt("l", 8, 1.0, mt, 0, lu, ci, rp, o["idiag"], bh, True)
        _, xk4 = oracle.trsv_kt("l", 4, 1.0, mt, 0, lu, ci, rp, o["idiag"], bh, True)
        for kid, nm, xref in ((-1, "auto: ref_trsv_l order, schedule chosen from the plan", xr),
                              (3, "kid 3: kt_trsv_l<512-bit> order (AVX-512 host), block kernel with run-time KT loops", xk8),
                              (1, "kid 1: kt_trsv_l<256-bit> order (AVX2 host), block kernel with run-time KT loops", xk4)):
            if kid not in kids:
                continue
            reps = 20 if kid < 0 else 10
            lp = timed_laps(pkg, lambda: pkg.dtrsv(pkg.OP_NONE, 1.0, At, dl, bdev, xdev, kid=kid), reps, 2)
            torch.cuda.synchronize()
            xg = xdev.cpu().numpy()
            _, lx = oracle.dcsrmv_special("tri", 0, 1.0, mt, mt, 1, 0, lu, ci, rp, o["idiag"], o["iurow"], xg, 0.0, np.zeros(mt))
            ms = float(np.median(lp))
            xr_k = xref
            res["schedules"].append({"schedule": nm, "kid": kid, "ms": round(ms, 5), "stats_ms": quartiles(lp),
                                     "us_per_level": round(ms * 1e3 / max(lv, 1), 4),
                                     "gflops": round((2.0 * nnz_l + mt) / ms / 1e6, 2),
                                     "roofline": roofline(ab, ms, note="bound by the dependency chain, not by HBM"),
                                     "bit_exact_vs_cpu": bool(np.array_equal(xg, xr_k)),
                                     "residual_inf": float(np.max(np.abs(lx - bh)) / np.max(np.abs(bh))),
                                     "max_componentwise_err_vs_ones": float(np.max(np.abs(xg - 1.0)))})
        return res

    run_leg("trsv", leg_trsv)

    # ---- aoclsparse_sp2m: C = A * A, full computation, host arrays in / host arrays out ----
    # The call is the reference's: operands are handles over host arrays, the result handle owns host arrays (export reads them).
    # ms = wall time of the whole call (analysis on the host, both kernel passes, the result over PCIe); kernels_ms = the two
    # SpGEMM passes alone (events around a second pair of calls would need the library's internals: taken from the phase trace).
    def leg_sp2m():
        import ctypes
        import oracle
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import standins
        L = pkg.lib()
        d = pkg.Descr()
        cases = [("5-pt Laplacian grid %d^2" % (300 if args.small else 1000),) + tuple(entry.laplace5(300 if args.small else 1000))]
        if not args.small:
            cases.append(("shell-like mesh, 100,000 rows",) + tuple(standins.shell_like(n=100000)))
        out = {"what": "aoclsparse_sp2m(A, A), full computation; host arrays in, host arrays out (PCIe inside the figure)", "cases": []}
        for title, mt, rp, ci, v in cases:
            A = pkg.Matrix(0, mt, mt, rp, ci, v)

            def run():
                C = ctypes.c_void_p()
                t = time.perf_counter()
                st = L.aoclsparse_sp2m(pkg.OP_NONE, d.h, A.h, pkg.OP_NONE, d.h, A.h, pkg.STAGE_FULL, ctypes.byref(C))
                dt = time.perf_counter() - t
                if st != 0:
                    raise RuntimeError("aoclsparse_sp2m: status %d" % st)
                return C, dt
            C, _ = run()
            L.aoclsparse_destroy(ctypes.byref(C))
            ts = []
            for k in range(5):
                C, dt = run()
                ts.append(dt)
                if k < 4:
                    L.aoclsparse_destroy(ctypes.byref(C))
            e = pkg.Matrix.from_handle(C).export()
            t = time.perf_counter()
            so, pc, ic, vc = oracle.dcsr2m(mt, mt, 0, rp, ci, v, 0, rp, ci, v)
            t_cpu = time.perf_counter() - t
            nnz_c = int(e["nnz"])
            by = 2 * (12 * len(v) + 4 * (mt + 1)) + 12 * nnz_c + 4 * (mt + 1)  # A read by both passes, C written once
            ms = sorted(ts)[len(ts) // 2] * 1e3
            out["cases"].append({"matrix": title, "m": mt, "nnz_a": int(len(v)), "nnz_c": nnz_c, "ms": round(ms, 3),
                                 "ms_all": [round(x * 1e3, 3) for x in ts], "algorithmic_bytes": by,
                                 "algorithmic_GBs_incl_pcie": round(by / ms / 1e6, 1),
                                 "cpu_port_1_thread_ms": round(t_cpu * 1e3, 2), "speedup_vs_cpu_port": round(t_cpu * 1e3 / ms, 2),
                                 "bit_exact": bool(so == 0 and np.array_equal(e["row_ptr"], pc) and np.array_equal(e["col_ind"], ic)
                                                   and np.array_equal(e["val"], vc))})
        return out

    run_leg("sp2m", leg_sp2m)

    # ---- in-library multi-device csrmm: ONE process over every visible GPU (aoclsparse_mi355_dcsrmm_multi_slabs) ----
    # Run as a CHILD process with a timeout: a problem on a multi-GPU node (first contact with N devices happens in the
    # driver's runs) must not take this line down.  On a one-GPU box the same control flow runs with two slots on device 0.
    def leg_inlib_multi():
        import subprocess
        cmd = [sys.executable, os.path.join(ROOT, "tools", "multi_check.py"), "--cols", str(args.mm_cols), "--grid", str(args.mm_grid)]
        cmd += ["--devices", str(min(ndev, 8))] if ndev > 1 else ["--devices", "2", "--same-device"]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not lines:
            return {"error": "multi_check.py exit %d: %s" % (r.returncode, r.stderr[-400:])}
        return json.loads(lines[-1])

    run_leg("inlib_multi", leg_inlib_multi)

    # ---- CPU baseline: the oracle on this box's host cores, bounded sample ----
    def leg_cpu():
        import oracle
        model, logical, phys = cpu_model, cpu_logical, cpu_physical
        res = {}
        for label, nthr in (("all_physical_cores", max(1, min(phys, oracle.max_threads()))), ("one_thread", 1)):
            _, s1, _ = oracle.dcsrmv_bench(-1, 0, m, m, nnz, val, col_ind, row_ptr, xh, nthr, 2)
            one = float(np.min(s1))
            passes = max(3, min(100, int(args.cpu_seconds / max(one, 1e-4))))
            stc, secs, yc = oracle.dcsrmv_bench(-1, 0, m, m, nnz, val, col_ind, row_ptr, xh, nthr, passes)
            med = float(np.median(secs))
            res[label] = {"threads": nthr, "passes": passes, "median_ms": round(med * 1e3, 4),
                          "stats_ms": quartiles([s * 1e3 for s in secs]), "gflops": round(flops / med / 1e9, 3),
                          "gbs": round(abytes / med / 1e9, 2),
                          "bit_exact_vs_gpu": bool(np.array_equal(yc, y.cpu().numpy()))}
        unpin()
        a = res["all_physical_cores"]
        return {"value": a["gflops"], "unit": "GFLOP/s", "cores": a["threads"], "kind": "port",
                "sample": "%d passes (all physical cores) + %d passes (one thread) of the same %dx%d-grid Laplacian SpMV "
                          "with the oracle: reference dispatch rule (nnz <= 10 m -> ref_csrmv_gn scalar order), OpenMP "
                          "static row split, arrays first-touched by the threads that read them, OMP_PROC_BIND=%s "
                          "OMP_PLACES=%s; median per pass"
                          % (a["passes"], res["one_thread"]["passes"], g, g, os.environ.get("OMP_PROC_BIND"),
                             os.environ.get("OMP_PLACES")),
                "gbs": a["gbs"], "cpu_model": model, "logical_cpus": logical, "physical_cores": phys,
                "omp_max_threads": oracle.max_threads(), "all_physical_cores": a, "one_thread": res["one_thread"],
                "bit_exact_vs_gpu": a["bit_exact_vs_gpu"]}

    if "cpu" in legs and rank == 0 and world == 1:
        try:
            out["cpu_baseline"] = leg_cpu()
        except Exception as e:
            out["cpu_baseline"] = {"error": "%s: %s" % (type(e).__name__, e)}
    elif rank == 0:
        out["cpu_baseline"] = None

    if rank == 0:
        for k in ("l100", "spmv_row_sharded", "sp2m_row_sharded", "csrmm_sharded_col", "csrmm_sharded_row", "csrmm_sharded_bell"):
            if k in legs_out:
                out[k] = legs_out.pop(k)
        out["legs"] = legs_out
        out = _sanitize(out)
        record_path = None
        if args.record:
            try:
                with open(args.record, "w") as f:
                    json.dump(out, f, allow_nan=False, indent=1)
                record_path = os.path.relpath(args.record, ROOT)
            except OSError as e:  # a read-only tree must not cost the record
                print("bench.py: could not write %s: %s" % (args.record, e), file=sys.stderr)
        sys.stdout.flush()
        print(compact_record(out, record_path), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
