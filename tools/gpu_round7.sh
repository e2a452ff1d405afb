#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -12 > gpurun_out/t_all_gpu.txt
cat gpurun_out/t_all_gpu.txt
python tools/bench_extra.py --what pcie 2>/dev/null | cut -c1-260 | tee gpurun_out/pcie_after_plan_fix.jsonl
AOCLSPARSE_MI355_PIPELINED_COPY=1 python tools/bench_extra.py --what pcie 2>/dev/null | cut -c1-260 | sed 's/"kind": "pcie-inclusive"/"kind": "pcie-inclusive, AOCLSPARSE_MI355_PIPELINED_COPY=1"/' | tee -a gpurun_out/pcie_after_plan_fix.jsonl
( time python bench.py > gpurun_out/bench_try5.json 2> gpurun_out/bench_try5.err ) 2>&1 | tail -4
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_try5.json').read().strip().splitlines()[-1])
print("value", d["value"], "frac", d["roofline"]["frac"])
print(d["legs"]["mix"]["merge_path_selection"])
print({k:v for k,v in d["legs"].items() if "error" in str(v)[:200]})
PY
