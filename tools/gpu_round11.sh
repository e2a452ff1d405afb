#!/bin/bash
mkdir -p gpurun_out
{
for sh in 0 8; do
echo "== shape $sh"
AOCLSPARSE_MI355_TRSV_BLK_SHAPE=$sh timeout 300 python tools/trsv_trace.py 2>/dev/null | cut -c 190-900
AOCLSPARSE_MI355_TRSV_BLK_SHAPE=$sh timeout 300 python tools/exp_trsv.py | grep shell | head -1
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/trsv_exp5.txt
timeout 900 python -m pytest tests/ -x -q -m gpu -k "trsv or trsm or symgs or ilu or itsol or sorv or csrsv" 2>&1 | tail -5
