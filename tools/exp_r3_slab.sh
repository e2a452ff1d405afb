#!/bin/bash
# round 3: what bounds the 32-column row-major slab (csrmm_tile_kernel)?  tile size of the row blocks, occupancy (LDS pad)
cd ${GRAFT_REPO_ROOT:-.}
for mode in 0 1; do
for tile in 512 1024 2048; do
  for pad in 0 16384 32768 65536; do
    echo -n "overwrite=$mode tile=$tile pad=$pad: "
    AOCLSPARSE_MI355_CSRMM_BETA0_OVERWRITE=$mode AOCLSPARSE_MI355_SPMV_TILE=$tile AOCLSPARSE_MI355_EXP_TILE_PAD=$pad python tools/exp_mm_lap.py 32 row | grep -o '"ms": [0-9.]*'
  done
done
done
