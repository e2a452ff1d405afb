#!/bin/bash
# One GPU-box pass that produces everything profiles/<round>/ holds.  Usage: bash tools/profile_round.sh <outdir>
# (run through gpurun; rocprofv3 gets the python program itself after `--`, counters in passes of their own:
# --pmc never together with --stats / sys-trace)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-prof}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PY=/usr/bin/python3
# 1. the default bench line, exactly as the driver runs it
$PY $R/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
cp $R/bench_legs.json $OUT/bench_legs_default.json   # the full report of the same run (stdout carries the short record only)
# 2. headline kernel alone (same timed region 1: every product from a flushed Infinity Cache; --cold-only drops the back-to-back
#    region, so EVERY launch of the kernel in these runs is a cold one and the --stats average is the cold average): kernel trace
#    + stats, then HBM traffic
SPMV="--steps 100 --warmup 10 --legs none --cold-only"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o spmv -- $PY $R/bench.py $SPMV > $OUT/bench_under_rocprof_trace.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o spmv -- $PY $R/bench.py $SPMV > $OUT/bench_pmc_fetch.json 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o spmv -- $PY $R/bench.py $SPMV > $OUT/bench_pmc_write.json 2> $OUT/pmc_write.err
# 3. the other configs' kernels: csrmm (both layouts, 256 and 32 columns), the mix, TRSV, raw dcsrmv -- stats, then traffic
LEGS="--steps 20 --warmup 3 --legs dcsrmv_csr_adaptive,headline_twins,mix,csrmm,trsv"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/legs_trace -o legs -- $PY $R/bench.py $LEGS > $OUT/bench_legs_under_rocprof.json 2> $OUT/legs_trace.err
cp $R/bench_legs.json $OUT/bench_legs_under_rocprof_full.json
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/legs_fetch -o legs -- $PY $R/bench.py $LEGS > /dev/null 2> $OUT/legs_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/legs_write -o legs -- $PY $R/bench.py $LEGS > /dev/null 2> $OUT/legs_write.err
# 3b. secondary measurements (stand-in csrmm, TRSV schedules, CG, section-8f rows, PCIe-inclusive rates with and without
#     the pipelined pinned copies)
cd $R
$PY tools/bench_extra.py > $OUT/extra_measurements.jsonl 2> $OUT/extra.err
cd /tmp
# 4. summaries (small text files: these are what gets committed under profiles/<round>/)
cd $R
for d in trace legs_trace; do
  f=$(ls $OUT/$d/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && cp $f $OUT/${d}_kernel_stats.csv
done
[ -f $OUT/trace_kernel_stats.csv ] && cp $OUT/trace_kernel_stats.csv $OUT/spmv_g4096_cold_kernel_stats.csv
{
  for c in FETCH_SIZE WRITE_SIZE; do
    d=pmc_$(echo $c | cut -d_ -f1 | tr A-Z a-z)
    echo "## headline, $c (KB per dispatch; FETCH_SIZE x2 on gfx950)"
    $PY tools/pmc_summary.py "$OUT/$d/*counter_collection.csv" $c
    d=legs_$(echo $c | cut -d_ -f1 | tr A-Z a-z)
    echo "## legs, $c"
    $PY tools/pmc_summary.py "$OUT/$d/*counter_collection.csv" $c
  done
} > $OUT/pmc_summary.txt 2>&1
ls $OUT | head -40
