#!/bin/bash
# round 3 (experiment build): row-per-wave csrmm with C read, old kernel vs the C-first / one-batch kernel, same box
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2 3; do for rw in 0 1 2 3; do  # (HISTORICAL: the switch existed in the experiment build only) for n in 256 128; do
  echo -n "rep=$rep new_kernel=$rw n=$n: "; AOCLSPARSE_MI355_EXP_RW=$rw python tools/exp_mm_lap.py $n row 2>/dev/null | grep -o '"ms": [0-9.]*'
done; done; done
