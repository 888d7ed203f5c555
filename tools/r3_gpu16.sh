#!/bin/bash
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/r3_gpu_tests.log 2>&1; tail -4 gpurun_out/r3_gpu_tests.log
timeout 900 python bench.py --dp-selftest --no-variants --no-cpu-baseline > gpurun_out/r3_dpself15.json 2> gpurun_out/r3_dpself15.err; python - <<'PY'
import json
try:
    L=open("gpurun_out/r3_dpself15.json").read().strip().splitlines(); print(len(L),"stdout lines")
    d=json.loads(L[-1]); print("dpself", d["value"], json.dumps(d.get("data_parallel")), d["host_enqueue_ms_per_step"], d["host_cpu_ms_per_step"])
except Exception as e: print("dpself failed", e, open("gpurun_out/r3_dpself15.err").read()[-1500:])
PY
ls /tmp/ytvln_rccl_*.log 2>/dev/null | head -2; head -30 /tmp/ytvln_rccl_*.log 2>/dev/null | cut -c1-200
