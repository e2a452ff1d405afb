#!/bin/bash
# round 3: slab kernel, C requested before the B rows (strict beta = 0 / beta != 0) + HBM-side traffic of both modes
cd ${GRAFT_REPO_ROOT:-.}
R=$(pwd)
for mode in 0 1; do for tile in 512 1024; do
  echo -n "overwrite=$mode tile=$tile row: "; AOCLSPARSE_MI355_CSRMM_BETA0_OVERWRITE=$mode AOCLSPARSE_MI355_SPMV_TILE=$tile python tools/exp_mm_lap.py 32 row 2>/dev/null | grep -o '"ms": [0-9.]*'
done; done
for mode in 0 1; do
  echo -n "overwrite=$mode col: "; AOCLSPARSE_MI355_CSRMM_BETA0_OVERWRITE=$mode python tools/exp_mm_lap.py 32 col 2>/dev/null | grep -o '"ms": [0-9.]*'
done
cd /tmp && export TMPDIR=/tmp
for mode in 0 1; do for c in FETCH_SIZE WRITE_SIZE; do
  export AOCLSPARSE_MI355_CSRMM_BETA0_OVERWRITE=$mode
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_${mode}_$c -o slab -- /usr/bin/python3 $R/tools/exp_mm_lap.py 32 row > /dev/null 2>&1
  echo "== overwrite=$mode $c (KB per dispatch of the slab kernel)"
  /usr/bin/python3 $R/tools/pmc_summary.py "/tmp/pmc_${mode}_$c/*counter_collection.csv" $c 2>&1 | grep -i "tile\|csrmm" | head -3
done; done
