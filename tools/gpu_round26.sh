#!/bin/bash
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/ -q -m gpu 2>&1 | tail -5 | tee gpurun_out/pytest_gpu_r2d.txt
bash tools/profile_round.sh r2prof > gpurun_out/profile_round.log 2>&1
tail -2 gpurun_out/profile_round.log
