#!/bin/bash
mkdir -p gpurun_out
{
for sf in 2 3; do
  AOCLSPARSE_MI355_TRSV_SYNCFREE=$sf timeout 200 python tools/exp_trsv.py
done
AOCLSPARSE_MI355_TRSV_SYNCFREE=3 AOCLSPARSE_MI355_TRSV_WAVES=8 timeout 200 python tools/exp_trsv.py
} 2>&1 | grep -v amdgpu.ids > gpurun_out/trsv_exp2.txt
cat gpurun_out/trsv_exp2.txt
for a in "300000 5 35 256" "500000 3 81 256"; do timeout 120 tools/bin/mfma_f64_probe $a; done > gpurun_out/mfma_probe.jsonl 2>&1
cat gpurun_out/mfma_probe.jsonl
for a in "2000 64" "8000 64" "700 64"; do timeout 120 tools/bin/latency_floor $a; done > gpurun_out/latency_floor.jsonl 2>&1
cat gpurun_out/latency_floor.jsonl
timeout 1500 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -15 > gpurun_out/t_all_gpu.txt
cat gpurun_out/t_all_gpu.txt
( time python bench.py > gpurun_out/bench_try4.json 2> gpurun_out/bench_try4.err ) 2>&1 | tail -4
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_try4.json').read().strip().splitlines()[-1])
print("value", d["value"], "frac", d["roofline"]["frac"])
cb=d["cpu_baseline"]; print("cpu", cb["value"], cb["cores"], cb["one_thread"]["gflops"])
print("adaptive", d["legs"]["dcsrmv_csr_adaptive"]["ms"], d["legs"]["dcsrmv_csr_adaptive"]["roofline"]["frac"])
for c in d["legs"]["csrmm"]["cases"]: print(c["layout"], c["ncols"], c["beta"], c["ms"], c["roofline"]["frac"], c["bit_exact_4_columns"])
for s in d["legs"]["trsv"]["schedules"]: print(s["schedule"], s["ms"], s["us_per_level"], s["bit_exact_vs_cpu"])
print({k:v for k,v in d["legs"].items() if "error" in str(v)[:200]})
PY
