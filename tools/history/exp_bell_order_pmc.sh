#!/bin/bash
# round 6: csrmm_bell_mfma_kernel, order of the block rows over the XCDs (AOCLSPARSE_MI355_BELL_XCD_CHUNK: 0 launch order, c chunks, -1 lattice
# sweep): time from tools/history/exp_bell.py, counters from --pmc passes of their own.  Usage (through gpurun):
# bash tools/exp_bell_order_pmc.sh <outdir> "<orders>" "<counters>" <edge>
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-bell_order}
ORD=${2:-"0 4 -1"}
CTRS=${3:-"FETCH_SIZE WRITE_SIZE"}
E=${4:-32}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PY=/usr/bin/python3
for v in $ORD; do
  export AOCLSPARSE_MI355_BELL_XCD_CHUNK=$v
  for rep in 1 2; do $PY $R/tools/history/exp_bell.py 256 $E 2>/dev/null | tail -1 | sed "s/^/order $v edge $E: /"; done
  for c in $CTRS; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/p${v}_$c -o x -- $PY $R/tools/history/exp_bell.py 256 $E > /dev/null 2> $OUT/p${v}_$c.err
    $PY $R/tools/pmc_summary.py "$OUT/p${v}_$c/*counter_collection.csv" $c | grep bell_mfma | sed "s/^/order $v edge $E: /" | sed "s/void mi355::csrmm_bell_mfma_kernel//; s/  */ /g"
  done
done 2>&1 | tee $OUT/bell_order_$E.txt
