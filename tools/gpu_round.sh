#!/bin/bash
# The usual GPU-box pass of a working session (run through gpurun): the GPU tier of the tests, then everything
# profiles/<round>/ holds (tools/profile_round.sh).  The one-off experiment scripts of round 2 were folded into the
# tools/exp_*.py / tools/*_trace.py programs they called; each evidence file under profiles/ names its command.
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/ -q -m gpu 2>&1 | tail -5 | tee gpurun_out/pytest_gpu.txt
bash tools/profile_round.sh ${1:-prof} > gpurun_out/profile_round.log 2>&1
tail -2 gpurun_out/profile_round.log
