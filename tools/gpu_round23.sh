#!/bin/bash
AOCLSPARSE_MI355_TIMING=1 timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sell" 2>&1 | grep -E "sell:|passed|failed|Error|assert" | sort | uniq -c | head -20
