#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for name in web-like circuit-like "web-like, local" "circuit-like, local"; do
  python tools/exp_sell_shared.py "$name" 2>&1 | grep -v amdgpu.ids
  for T in 16 32 64; do
    python tools/exp_r3_sellsigma.py long "$name" $T 2>&1 | grep -v amdgpu.ids
    for sigma in 1024 4096; do
      AOCLSPARSE_MI355_SELL=1 python tools/exp_r3_sellsigma.py short "$name" $T $sigma 2>&1 | grep -v amdgpu.ids
  done; done
done
