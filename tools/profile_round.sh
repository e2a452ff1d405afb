#!/bin/bash
# One GPU-box pass that produces everything profiles/<round>/ holds.  Usage: bash tools/profile_round.sh <outdir>
# (run through gpurun; rocprofv3 gets the python program itself after `--`, counters in passes of their own)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-prof}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
SPMV="--steps 50 --warmup 5 --no-cpu --no-l100 --no-csrmm"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py $SPMV > $OUT/bench_under_rocprof_trace.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py $SPMV > $OUT/bench_pmc_fetch.json 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py $SPMV > $OUT/bench_pmc_write.json 2> $OUT/pmc_write.err
python3 $R/tools/bench_extra.py > $OUT/extra_measurements.jsonl 2> $OUT/extra.err
ls -R $OUT | head -40
