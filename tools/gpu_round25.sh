#!/bin/bash
run() { timeout 600 python bench.py --legs dcsrmv_csr_adaptive,trsv --steps 50 --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', d['legs']['dcsrmv_csr_adaptive']['ms'], d['legs']['trsv']['schedules'][0]['ms'])"; }
run new; AOCLSPARSE_MI355_LIB=$PWD/tools/bin/libaoclsparse_mi355_old.so run old; run new
timeout 300 python tools/trsv_trace.py 2>&1 | grep -o '"total_us": [0-9.]*\|"work_us_q[^]]*\]'
timeout 300 python tools/spmv_trace.py 2>&1 | grep -o '"matrix": "[a-z-]*"\|"kernel_span_us[^,]*'
timeout 1200 python -m pytest tests/ -x -q -m gpu -k "trsv or trsm or mv or spmv or mix or heavy or block" 2>&1 | tail -3
