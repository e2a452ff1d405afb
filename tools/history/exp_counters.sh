#!/bin/bash
# Counter sweep over one program: bash tools/exp_counters.sh <tag> <script under the repo> <args...>  (KERN=<regex of kernel names>); one rocprofv3 --pmc pass per counter group
# (each pass under its own timeout: the TA_* group aborted inside rocprofv3 on this image and sat there until gpurun's limit)
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; PROG=$2; shift; shift
OUT=$R/gpurun_out/ctr_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PY=/usr/bin/python3
[ -f $R/gpurun_out/avail.txt ] || rocprofv3 --list-avail > $R/gpurun_out/avail.txt 2>&1
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_VALU" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM SQ_WAIT_ANY" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum" \
           "GRBM_GUI_ACTIVE TCC_BUSY_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum" \
           "MemUnitStalled MeanOccupancyPerCU TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/g$i -o x -- $PY $R/$PROG "$@" > /dev/null 2> $OUT/err_g$i.txt
  for c in $grp; do
    $PY $R/tools/pmc_summary.py "$OUT/g$i/*counter_collection.csv" $c 2>/dev/null | grep -E "${KERN:-csrmm}" | sed "s/^/$c  /"
  done
done
