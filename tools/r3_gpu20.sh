#!/bin/bash
mkdir -p gpurun_out
S=$(date +%s)
timeout 1800 python bench.py > gpurun_out/round3_bench.json 2> gpurun_out/round3_bench.err
E=$(date +%s); echo "default bench wall: $((E-S)) s, stdout lines: $(wc -l < gpurun_out/round3_bench.json)"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/round3_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["clock"].get("shader_ghz_under_load"), d["roofline"]["clock"].get("main_loop_cycle_frac"))
for k,v in d["variants"].items(): print(k, {x:v.get(x) for x in ("value","ms_per_step","model_frac_of_bf16_peak","error")})
print(d["cpu_baseline"]["value"])
PY
timeout 1500 python -m pytest tests/test_rccl_gpu.py -q -m gpu 2>&1 | grep -E "passed|failed"
