#!/bin/bash
mkdir -p gpurun_out/pmc_irr
R=$GRAFT_REPO_ROOT
B=tools/bin/csrmm_r2
{
for cfg in "1000 256 1000" "1000 128 1000"; do
  echo "=== $cfg"
  timeout 300 $B $cfg "R0,RE ,RS L64 R1 NB8 rowmap,diag D2,diag D3,copy simple"
done
} > gpurun_out/csrmm_r2_exp5.txt 2>&1
grep -v "^#" gpurun_out/csrmm_r2_exp5.txt
for a in "300000 5 35 256" "500000 3 81 256"; do timeout 120 tools/bin/mfma_f64_probe $a; done > gpurun_out/mfma_probe.jsonl 2>&1
cat gpurun_out/mfma_probe.jsonl
timeout 900 python tools/exp_arrow.py 2>&1 | grep -v amdgpu.ids > gpurun_out/merge_vs_adaptive.jsonl
cat gpurun_out/merge_vs_adaptive.jsonl
AOCLSPARSE_MI355_TRSV_SYNCFREE=3 timeout 200 python tools/exp_trsv.py 2>&1 | grep -v amdgpu.ids > gpurun_out/trsv_exp3.txt
cat gpurun_out/trsv_exp3.txt
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/pmc_irr/$c -o out --output-format csv -- /usr/bin/python3 $R/tools/exp_irregular.py web-like circuit-like > $R/gpurun_out/pmc_irr/$c.log 2>&1
done
cd $R
python3 tools/pmc_table.py gpurun_out/pmc_irr > gpurun_out/irregular_pmc.txt 2>&1
grep -A3 "csr_adaptive" gpurun_out/irregular_pmc.txt | head -20
