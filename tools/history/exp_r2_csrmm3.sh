#!/bin/bash
mkdir -p gpurun_out
B=tools/bin/csrmm_r2
F="R0,TL,RT,RS L16 R1 NB8 rowmap,RS L64 R1 NB8 rowmap,diag D2,copy simple"
{
for cfg in "1000 256 1000" "1000 32 1000" "1000 128 1000" "1000 64 1000" "1000 256 100"; do
  echo "=== $cfg"
  timeout 300 $B $cfg "$F"
done
} > gpurun_out/csrmm_r2_exp3.txt 2>&1
grep -v "^#" gpurun_out/csrmm_r2_exp3.txt
