#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_trsv_blocks.py -x -q -m gpu 2>&1 | tail -5
timeout 900 python -m pytest tests/ -x -q -m gpu -k "trsm or trsv or symgs or ilu" 2>&1 | tail -4
{
timeout 600 python tools/exp_trsm.py
AOCLSPARSE_MI355_TRSV_BLOCKS=0 timeout 600 python tools/exp_trsm.py
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/trsm_exp.txt
