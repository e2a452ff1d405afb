#!/bin/bash
mkdir -p gpurun_out
{
timeout 600 python tools/exp_trsv.py --all
AOCLSPARSE_MI355_TRSV_BLOCKS=0 timeout 600 python tools/exp_trsv.py --all | grep '"kid": -1'
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/trsv_exp6.txt
timeout 900 python -m pytest tests/ -x -q -m gpu -k "trsv or trsm or symgs or ilu or itsol or sorv or csrsv" 2>&1 | tail -5
