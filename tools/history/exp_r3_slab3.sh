#!/bin/bash
# HISTORICAL: the AOCLSPARSE_MI355_EXP_* / _STRIP_* switches this script sets existed only in the experiment builds whose
# output is kept under profiles/; the library no longer reads them (the winning setting is compiled in).
# round 3: slab kernel shape sweep (rows in flight x loads per row), both beta = 0 modes, same box, interleaved twice
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do
for mode in 0 1; do for shape in 28 26 36 46 48 18; do for tile in 512 1024; do
  echo -n "rep=$rep overwrite=$mode shape=$shape tile=$tile: "; AOCLSPARSE_MI355_EXP_TILE_SHAPE=$shape AOCLSPARSE_MI355_CSRMM_BETA0_OVERWRITE=$mode AOCLSPARSE_MI355_SPMV_TILE=$tile python tools/exp_mm_lap.py 32 row 2>/dev/null | grep -o '"ms": [0-9.]*'
done; done; done
done
