#!/bin/bash
mkdir -p gpurun_out
{
timeout 300 python tools/exp_trsv.py
AOCLSPARSE_MI355_TRSV_BLOCKS=0 timeout 300 python tools/exp_trsv.py --small
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/trsv_exp4.txt
timeout 900 python -m pytest tests/ -x -q -m gpu -k "trsv or trsm or symgs or ilu or itsol or sorv or csrsv or bench" 2>&1 | tail -8
