#!/bin/bash
# Row-major csrmm on the 1000^2 Laplacian: strip order on / off, time and L2-miss traffic (FETCH_SIZE, KB, x2 on gfx950).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/strips
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PY=/usr/bin/python3
for n in ${NS:-256 128}; do
for st in 1 0; do
  export AOCLSPARSE_MI355_CSRMM_STRIPS=$st
  echo "== n=$n strips=$st"
  $PY $R/tools/exp_mm_lap.py $n
  $PY $R/tools/exp_mm_lap.py $n
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/f_${n}_$st -o x -- $PY $R/tools/exp_mm_lap.py $n > /dev/null 2> $OUT/err_${n}_$st.txt
  $PY $R/tools/pmc_summary.py "$OUT/f_${n}_$st/*counter_collection.csv" FETCH_SIZE | grep csrmm
done
done
