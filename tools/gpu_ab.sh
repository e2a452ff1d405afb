#!/bin/bash
for r in 1 0 1 0; do AOCLSPARSE_MI355_CSRMM_RUNS=$r python3 tools/exp_mm_lap.py 256 2>/dev/null; done
timeout 900 python -m pytest tests/ -x -q -m gpu -k "csrmm" 2>&1 | tail -3
