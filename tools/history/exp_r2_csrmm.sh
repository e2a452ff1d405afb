#!/bin/bash
# round-2 csrmm experiments (diagnostic): tools/bin/csrmm_r2 over sizes / bands; output under gpurun_out/
mkdir -p gpurun_out
B=tools/bin/csrmm_r2
for cfg in "1000 256 1000" "1000 256 100" "1000 256 10" "1000 256 5000" "1000 32 1000" "1000 32 100" "1000 64 1000" "1000 128 1000"; do
  echo "=== $cfg" 
  timeout 300 $B $cfg
done > gpurun_out/csrmm_r2_exp1.txt 2>&1
tail -5 gpurun_out/csrmm_r2_exp1.txt
