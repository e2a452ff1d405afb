#!/bin/bash
mkdir -p gpurun_out
{
AOCLSPARSE_MI355_TIMING=1 timeout 600 python tools/exp_ilu.py
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ilu_exp.txt
