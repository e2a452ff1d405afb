#!/bin/bash
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -m gpu -k "sell or mix or mv or spmv or l100" 2>&1 | tail -4
for i in 1 2; do timeout 600 python bench.py --legs none 2>/dev/null | cut -c1-160; done
timeout 600 python tools/exp_sell_shared.py 2>/dev/null
