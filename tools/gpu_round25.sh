#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "row_runs" 2>&1 | tail -12
