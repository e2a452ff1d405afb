#!/bin/bash
# round 3 (experiment build): slices per wavefront x wavefronts per workgroup of sell_mv_short_kernel on the headline workload
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2 3; do for cfg in 98 99 0 11 12 14 21 24; do
  echo -n "rep=$rep NS*10+WAVES=$cfg: "; AOCLSPARSE_MI355_EXP_SHORT=$cfg python bench.py --legs none --steps 100 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'], d['parity']['bit_exact'])"
done; done
