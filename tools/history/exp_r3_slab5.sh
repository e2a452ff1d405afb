#!/bin/bash
# HISTORICAL: the AOCLSPARSE_MI355_EXP_* / _STRIP_* switches this script sets existed only in the experiment builds whose
# output is kept under profiles/; the library no longer reads them (the winning setting is compiled in).
# round 3: slab kernel occupancy (min waves per SIMD through launch bounds) x load slots, both beta = 0 modes
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do
for mode in 0 1; do for nb5 in 0 1; do for w in 0 5 6; do
  echo -n "rep=$rep overwrite=$mode nb5=$nb5 minw=$w: "; AOCLSPARSE_MI355_EXP_TILE_W=$w AOCLSPARSE_MI355_EXP_TILE_NB5=$nb5 AOCLSPARSE_MI355_CSRMM_BETA0_OVERWRITE=$mode python tools/exp_mm_lap.py 32 row 2>/dev/null | grep -o '"ms": [0-9.]*'
done; done; done
done
