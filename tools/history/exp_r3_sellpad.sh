#!/bin/bash
# round 3: the unstructured mesh variants under SELL-64 forced (AOCLSPARSE_MI355_SELL=1: padding budget ignored) vs the default choice
cd ${GRAFT_REPO_ROOT:-.}
for mode in default 1; do
  if [ $mode = default ]; then unset AOCLSPARSE_MI355_SELL; else export AOCLSPARSE_MI355_SELL=$mode; fi
  echo "== AOCLSPARSE_MI355_SELL=$mode"
  python tools/exp_sell_shared.py "flan-like, unstructured" "shell-like, unstructured" 2>&1 | grep -v amdgpu.ids
done
