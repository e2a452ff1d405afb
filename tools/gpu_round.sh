#!/bin/bash
# one GPU session: new tests, csrmm regression tests, default bench line
mkdir -p gpurun_out
python -m pytest tests/test_gpu_configs.py -x -q 2>&1 | tail -25 > gpurun_out/t_configs.txt
cat gpurun_out/t_configs.txt
python -m pytest tests/test_gpu_parity.py -x -q -k "csrmm or trsv or samples" 2>&1 | tail -8 > gpurun_out/t_parity_sub.txt
cat gpurun_out/t_parity_sub.txt
( time python bench.py > gpurun_out/bench_try2.json 2> gpurun_out/bench_try2.err ) 2>&1 | tail -4
tail -c 300 gpurun_out/bench_try2.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_try2.json').read().strip().splitlines()[-1])
print("value", d["value"], "frac", d["roofline"]["frac"], "stats", d["stats"])
cb=d["cpu_baseline"]; print("cpu", cb["value"], cb["cores"], cb["cpu_model"], cb["physical_cores"], cb["one_thread"]["gflops"])
for c in d["legs"]["csrmm"]["cases"]: print(c["layout"], c["ncols"], c["beta"], c["ms"], c["roofline"]["frac"], c["bit_exact_4_columns"])
s=d["csrmm_sharded"]; print("sharded", s["layout"], s["tg_ms_device_median_max_over_ranks"], s["efficiency"], s["roofline_full"]["frac"])
for r in d["legs"]["mix"]["matrices"]: print(r["matrix"], r["kernel"], r["us"], r["roofline"]["frac"], r["cpu_all_cores_gflops"])
print({k:v for k,v in d["legs"].items() if "error" in str(v)[:200]})
print("leg seconds", {k:v.get("leg_seconds") for k,v in d["legs"].items()})
PY
