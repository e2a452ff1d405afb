#!/bin/bash
mkdir -p gpurun_out
B=tools/bin/csrmm_r2
{
for cfg in "1000 256 1000" "1000 128 1000"; do
  echo "=== $cfg"
  timeout 300 $B $cfg "R0,RE ell8 R1,RP ,diag D2,diag D3,copy simple"
done
} > gpurun_out/csrmm_r2_exp6.txt 2>&1
grep -v "^#" gpurun_out/csrmm_r2_exp6.txt
