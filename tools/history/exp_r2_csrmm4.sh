#!/bin/bash
mkdir -p gpurun_out
B=tools/bin/csrmm_r2
F="C0,CPAIR,CP U2 RL1 rowmap cc64,copy simple"
{
for cfg in "1000 256 1000" "1000 32 1000"; do
  echo "=== $cfg"
  timeout 300 $B $cfg "$F"
done
} > gpurun_out/csrmm_r2_exp4.txt 2>&1
grep -v "^#" gpurun_out/csrmm_r2_exp4.txt
timeout 900 python bench.py --steps 30 --warmup 5 > gpurun_out/bench_try1.json 2> gpurun_out/bench_try1.err
echo "bench rc=$?"; tail -c 600 gpurun_out/bench_try1.err; python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/bench_try1.json').read().strip().splitlines()[-1])
    print(json.dumps(d, indent=1)[:6000])
except Exception as e:
    print("parse failed", e)
PY
