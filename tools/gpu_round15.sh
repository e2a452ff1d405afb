#!/bin/bash
mkdir -p gpurun_out
{
timeout 600 python tools/exp_irregular_laps.py
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/spmv_trace6.txt
timeout 1200 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -x -q -m gpu -k "heavy or mix or csrmv or mv or spmv or plan" 2>&1 | tail -4
