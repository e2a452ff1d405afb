#!/bin/bash
# round 6: raw aoclsparse_dcsrmv on the 4096^2 Laplacian (csr_adaptive_kernel), launch order vs XCD-contiguous block order: time, fabric bytes,
# L2 hits / misses.  Counters in passes of their own (never together with --stats).  Usage (through gpurun): bash tools/exp_csrmv_xcd.sh <outdir>
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-csrmv_xcd}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PY=/usr/bin/python3
ARGS="--steps 20 --warmup 3 --cold-only --legs dcsrmv_csr_adaptive --record ''"
for ord in 0 1; do
  export AOCLSPARSE_MI355_SPMV_XCD_ORDER=$ord
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t$ord -o x -- $PY $R/bench.py $ARGS > /dev/null 2> $OUT/t$ord.err
  for c in FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/p${ord}_$c -o x -- $PY $R/bench.py $ARGS > /dev/null 2> $OUT/p${ord}_$c.err
  done
done
cd $R
{
  for ord in 0 1; do
    echo "## block order $ord (0 = launch order, 1 = XCD-contiguous)"
    grep "csr_adaptive" $OUT/t$ord/*kernel_stats.csv | cut -d, -f1-8 | cut -c1-200
    for c in FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum; do
      $PY tools/pmc_summary.py "$OUT/p${ord}_$c/*counter_collection.csv" $c | grep csr_adaptive
    done
  done
} > $OUT/csrmv_xcd_order_pmc.txt 2>&1
cat $OUT/csrmv_xcd_order_pmc.txt
