#!/bin/bash
# round 3: where does the row-group csrmm kernel spend its time on the flan-like stand-in (256 columns, row-major)?
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for name in flan-like shell-like; do python tools/exp_mm_standin.py $name 256 row; done
KERN=csrmm bash tools/exp_counters.sh flan256 tools/exp_mm_standin.py flan-like 256 row
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 120 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/ctr_flan256/$c -o x -- /usr/bin/python3 $R/tools/exp_mm_standin.py flan-like 256 row > /dev/null 2>&1
  /usr/bin/python3 $R/tools/pmc_summary.py "$R/gpurun_out/ctr_flan256/$c/*counter_collection.csv" $c | grep csrmm | sed "s/^/$c  /"
done
