#!/bin/bash
mkdir -p gpurun_out
{
for hf in 1 0 1 0; do
AOCLSPARSE_MI355_SPMV_HEAVY_FIRST=$hf timeout 600 python tools/exp_irregular_laps.py
done
timeout 600 python tools/spmv_trace.py
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/spmv_trace2.txt
