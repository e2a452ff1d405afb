#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -x -q -k "csrmm" 2>&1 | tail -12
python tools/bench_extra.py --what csrmm 2>/dev/null | grep -E "shell|flan" | cut -c1-330 | tee gpurun_out/csrmm_super_on.jsonl
AOCLSPARSE_MI355_CSRMM_SUPER=0 python tools/bench_extra.py --what csrmm 2>/dev/null | grep -E "shell|flan" | cut -c1-330 | tee gpurun_out/csrmm_super_off.jsonl
