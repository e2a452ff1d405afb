#!/bin/bash
mkdir -p gpurun_out
B=tools/bin/csrmm_r2
{
for cfg in "1000 256 1000" "1000 128 1000"; do
  echo "=== $cfg"
  timeout 300 $B $cfg "R0 shipped,RR reuse,diag D2,copy simple"
done
} 2>&1 | tee gpurun_out/csrmm_r2_exp9.txt
