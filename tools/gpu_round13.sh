#!/bin/bash
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/pytest_gpu_r2c.txt
{
echo "# tools/trsv_trace.py: block-kernel trace on the shell-like ILU(0) factor (L, unit), default gate 2 then gate 0"
timeout 300 python tools/trsv_trace.py
AOCLSPARSE_MI355_TRSV_GATE=0 timeout 300 python tools/trsv_trace.py
echo "# generic shape (8 rows x 16, row by row from LDS) forced on the same factor"
AOCLSPARSE_MI355_TRSV_BLK_SHAPE=8 timeout 300 python tools/trsv_trace.py
} 2>&1 | grep -v amdgpu.ids > gpurun_out/trsv_block_trace.txt
{
echo "# tools/exp_trsv.py --all : auto (kid -1) and kid 3, blocks on"
timeout 600 python tools/exp_trsv.py --all
echo "# AOCLSPARSE_MI355_TRSV_BLOCKS=0 (row-level schedules, auto rule)"
AOCLSPARSE_MI355_TRSV_BLOCKS=0 timeout 600 python tools/exp_trsv.py --all
echo "# AOCLSPARSE_MI355_TRSV_BLOCKS=0 AOCLSPARSE_MI355_TRSV_SYNCFREE=2"
AOCLSPARSE_MI355_TRSV_BLOCKS=0 AOCLSPARSE_MI355_TRSV_SYNCFREE=2 timeout 600 python tools/exp_trsv.py --all | grep '"kid": 3'
echo "# AOCLSPARSE_MI355_TRSV_BLOCKS=0 AOCLSPARSE_MI355_TRSV_SYNCFREE=3"
AOCLSPARSE_MI355_TRSV_BLOCKS=0 AOCLSPARSE_MI355_TRSV_SYNCFREE=3 timeout 600 python tools/exp_trsv.py --all | grep '"kid": 3'
} 2>&1 | grep -v amdgpu.ids > gpurun_out/trsv_schedules_r2c.txt
bash tools/profile_round.sh r2prof > gpurun_out/profile_round.log 2>&1
tail -3 gpurun_out/profile_round.log
