#!/bin/bash
timeout 1200 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -x -q -m gpu -k "heavy or mix or csrmv or mv or spmv or plan" 2>&1 | tail -6
