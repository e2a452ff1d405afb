#!/bin/bash
# round-2 csrmm diagnostics: chain-vs-cache experiment + PMC counters for the shipped row-major kernel
mkdir -p gpurun_out/pmc
B=tools/bin/csrmm_r2
F="R0,diag,copy,RS L64 R1 NB8 rowmap,RS L64 R2 NB8 rowmap nt"
{
for cfg in "1000 256 1000" "1000 256 10"; do
  echo "=== $cfg"
  timeout 300 $B $cfg "$F"
done
} > gpurun_out/csrmm_r2_exp2.txt 2>&1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "TCP_TCR_TCP_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_ACCESSES_sum" \
           "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum" \
           "TCC_BUSY_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TC_STALL_sum" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE TD_TD_BUSY_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace -d $R/gpurun_out/pmc/g$i -o out --output-format csv -- $R/$B 1000 256 1000 "R0,diag D1,diag D2,copy simple" 1 > $R/gpurun_out/pmc/g$i.log 2>&1
done
cd $R
python3 tools/pmc_table.py gpurun_out/pmc > gpurun_out/csrmm_r2_pmc.txt 2>&1
cat gpurun_out/csrmm_r2_exp2.txt | grep -v "^#"
cat gpurun_out/csrmm_r2_pmc.txt | head -60
