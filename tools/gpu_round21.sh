#!/bin/bash
mkdir -p gpurun_out
{
echo "# AOCLSPARSE_MI355_CSRMM_RG2=1 (default), tail batched"
timeout 900 python tools/bench_extra.py --what csrmm 2>&1 | grep like | grep '"n": 256' | grep row-major
} | tee gpurun_out/csrmm_rg2b.txt
timeout 900 python -m pytest tests/ -x -q -m gpu -k "csrmm" 2>&1 | tail -4
