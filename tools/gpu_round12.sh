#!/bin/bash
mkdir -p gpurun_out
{
for sf in 3 2; do
AOCLSPARSE_MI355_TRSV_BLOCKS=0 AOCLSPARSE_MI355_TRSV_SYNCFREE=$sf timeout 600 python tools/exp_trsv.py --all | grep '"kid": 3'
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/trsv_exp7.txt
