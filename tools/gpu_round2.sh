#!/bin/bash
mkdir -p gpurun_out
{
for sf in 2 3; do for wv in 16 8 4; do
  [ $sf = 2 ] && [ $wv != 16 ] && continue
  AOCLSPARSE_MI355_TRSV_SYNCFREE=$sf AOCLSPARSE_MI355_TRSV_WAVES=$wv timeout 600 python tools/exp_trsv.py
done; done
} > gpurun_out/trsv_exp1.txt 2>&1
cat gpurun_out/trsv_exp1.txt | grep -v amdgpu.ids
python -m pytest tests/test_gpu_configs.py -x -q 2>&1 | tail -25 > gpurun_out/t_configs.txt
cat gpurun_out/t_configs.txt
python -m pytest tests/test_gpu_parity.py -x -q -k "csrmm or trsv or trsm or symgs or ilu" 2>&1 | tail -8 > gpurun_out/t_parity_sub.txt
cat gpurun_out/t_parity_sub.txt
( time python bench.py > gpurun_out/bench_try3.json 2> gpurun_out/bench_try3.err ) 2>&1 | tail -4
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_try3.json').read().strip().splitlines()[-1])
print("value", d["value"], "frac", d["roofline"]["frac"])
cb=d["cpu_baseline"]; print("cpu", cb["value"], cb["cores"], cb["cpu_model"], cb["physical_cores"], cb["logical_cpus"], cb["one_thread"]["gflops"])
for c in d["legs"]["csrmm"]["cases"]: print(c["layout"], c["ncols"], c["beta"], c["ms"], c["roofline"]["frac"], c["bit_exact_4_columns"])
s=d["csrmm_sharded"]; print("sharded", s["layout"], s["tg_ms_device_median_max_over_ranks"], s["efficiency"], s["roofline_full"]["frac"])
for r in d["legs"]["mix"]["matrices"]: print(r["matrix"], r["kernel"], r["us"], r["roofline"]["frac"], r["cpu_all_cores_gflops"])
for s in d["legs"]["trsv"]["schedules"]: print(s["schedule"], s["ms"], s["us_per_level"], s["bit_exact_vs_cpu"])
print({k:v for k,v in d["legs"].items() if "error" in str(v)[:200]})
PY
