#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "graph or bench_single" 2>&1 | tail -15
