#!/bin/bash
# times, then fabric-side bytes (FETCH_SIZE x2 on gfx950 + WRITE_SIZE) per launch of the row-block kernel
cd ${GRAFT_REPO_ROOT:-.}; R=$(pwd); cd /tmp; export TMPDIR=/tmp
for x in 0 1; do
  export AOCLSPARSE_MI355_XCD_ORDER=$x
  /usr/bin/python3 $R/tools/exp_irregular_r3.py 2>/dev/null
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/irr_${x}_$c -o irr -- /usr/bin/python3 $R/tools/exp_irregular_r3.py > /dev/null 2>&1
    echo "== XCD_ORDER=$x $c (KB per dispatch, by grid size; FETCH_SIZE x2 on gfx950)"
    /usr/bin/python3 $R/tools/pmc_summary.py "/tmp/irr_${x}_$c/*counter_collection.csv" $c 2>&1 | grep -i "adaptive" | head -8
  done
done
