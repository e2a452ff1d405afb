#!/bin/bash
# HISTORICAL: the AOCLSPARSE_MI355_EXP_* / _STRIP_* switches this script sets existed only in the experiment builds whose
# output is kept under profiles/; the library no longer reads them (the winning setting is compiled in).
# Row-major csrmm on the 1000^2 Laplacian, 256 columns: strip width / band lines per workgroup group, time and L2-miss
# traffic (FETCH_SIZE, KB, x2 on gfx950).  CASES="rows:qgroup ..." (0:0 = strips off)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/strips
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PY=/usr/bin/python3
n=${N:-256}
for c in ${CASES:-128:1 0:0}; do
  rows=${c%%:*}; qg=${c##*:}
  if [ "$rows" = 0 ]; then export AOCLSPARSE_MI355_CSRMM_STRIPS=0; else export AOCLSPARSE_MI355_CSRMM_STRIPS=1 AOCLSPARSE_MI355_CSRMM_STRIP_ROWS=$rows AOCLSPARSE_MI355_CSRMM_STRIP_QGROUP=$qg; fi
  echo "== n=$n strip rows=$rows qgroup=$qg"
  $PY $R/tools/exp_mm_lap.py $n 2>/dev/null
  $PY $R/tools/exp_mm_lap.py $n 2>/dev/null
  timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/f_${n}_${rows}_$qg -o x -- $PY $R/tools/exp_mm_lap.py $n > /dev/null 2> $OUT/err_${n}_${rows}_$qg.txt
  $PY $R/tools/pmc_summary.py "$OUT/f_${n}_${rows}_$qg/*counter_collection.csv" FETCH_SIZE | grep csrmm
done
