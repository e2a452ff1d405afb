#!/bin/bash
mkdir -p gpurun_out
{
python tools/exp_strict.py
AOCLSPARSE_MI355_STRICT_LONG=8192 python tools/exp_strict.py
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/strict_long.jsonl
