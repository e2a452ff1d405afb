#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_rg
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SMEM" \
           "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum" \
           "TCC_BUSY_sum TCC_EA0_RDREQ_sum TA_TA_BUSY_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "GRBM_GUI_ACTIVE TD_TD_BUSY_sum TCP_GATE_EN1_sum TCP_TD_TCP_STALL_CYCLES_sum" \
           "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  for mat in shell-like flan-like; do
  timeout 300 rocprofv3 --pmc $grp --kernel-trace -d $R/gpurun_out/pmc_rg/$mat/g$i -o out --output-format csv -- /usr/bin/python3 $R/tools/exp_mm_standin.py $mat 256 > $R/gpurun_out/pmc_rg/$mat.g$i.log 2>&1
  done
done
cd $R
for mat in shell-like flan-like; do echo "#### $mat"; python3 tools/pmc_table.py gpurun_out/pmc_rg/$mat | grep -A60 "rowgroup2"; done > gpurun_out/csrmm_rg2_pmc.txt 2>&1
rm -rf gpurun_out/pmc_rg/*/g*/*/*.csv.bak
head -120 gpurun_out/csrmm_rg2_pmc.txt
