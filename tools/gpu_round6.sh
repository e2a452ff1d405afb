#!/bin/bash
mkdir -p gpurun_out
timeout 300 tools/bin/h2d_probe > gpurun_out/h2d_probe.jsonl 2>&1
cat gpurun_out/h2d_probe.jsonl
python -m pytest tests/test_gpu_configs.py -x -q -k "pinned_ring or merge_path or mv_triangular or plan_cache" 2>&1 | tail -8
python tools/bench_extra.py --what pcie 2>/dev/null | cut -c1-200
AOCLSPARSE_MI355_COPY_THREADS=16 python tools/bench_extra.py --what pcie 2>/dev/null | cut -c1-200
