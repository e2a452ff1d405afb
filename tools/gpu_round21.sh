#!/bin/bash
mkdir -p gpurun_out
{
echo "# AOCLSPARSE_MI355_CSRMM_RG2=1 (default)"
timeout 900 python tools/bench_extra.py --what csrmm 2>&1 | grep like
echo "# AOCLSPARSE_MI355_CSRMM_RG2=0"
AOCLSPARSE_MI355_CSRMM_RG2=0 timeout 900 python tools/bench_extra.py --what csrmm 2>&1 | grep like
} | tee gpurun_out/csrmm_rg2.txt
timeout 900 python -m pytest tests/ -x -q -m gpu -k "csrmm" 2>&1 | tail -4
