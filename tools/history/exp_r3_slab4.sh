#!/bin/bash
# HISTORICAL: the AOCLSPARSE_MI355_EXP_* / _STRIP_* switches this script sets existed only in the experiment builds whose
# output is kept under profiles/; the library no longer reads them (the winning setting is compiled in).
# round 3: column-major slab (csrmm_colpair_kernel): entries cached x columns per step, both beta = 0 modes; + the row-major
# slab with its fixed shape; + 256 columns both layouts (nothing else may regress)
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do
for mode in 0 1; do for shape in 84 82 64 62; do
  echo -n "rep=$rep overwrite=$mode cp_shape=$shape col32: "; AOCLSPARSE_MI355_EXP_CP_SHAPE=$shape AOCLSPARSE_MI355_CSRMM_BETA0_OVERWRITE=$mode python tools/exp_mm_lap.py 32 col 2>/dev/null | grep -o '"ms": [0-9.]*'
done; done
done
for mode in 0 1; do
  echo -n "overwrite=$mode row32: "; AOCLSPARSE_MI355_CSRMM_BETA0_OVERWRITE=$mode python tools/exp_mm_lap.py 32 row 2>/dev/null | grep -o '"ms": [0-9.]*'
  for shape in 84 64; do
  echo -n "overwrite=$mode cp_shape=$shape col256: "; AOCLSPARSE_MI355_EXP_CP_SHAPE=$shape AOCLSPARSE_MI355_CSRMM_BETA0_OVERWRITE=$mode python tools/exp_mm_lap.py 256 col 2>/dev/null | grep -o '"ms": [0-9.]*'
  done
  echo -n "overwrite=$mode row256: "; AOCLSPARSE_MI355_CSRMM_BETA0_OVERWRITE=$mode python tools/exp_mm_lap.py 256 row 2>/dev/null | grep -o '"ms": [0-9.]*'
done
