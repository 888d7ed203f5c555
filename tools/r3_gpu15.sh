#!/bin/bash
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r3_gpu_tests.log 2>&1; tail -5 gpurun_out/r3_gpu_tests.log
timeout 900 python bench.py --no-variants --no-cpu-baseline > gpurun_out/r3_bench15.json 2> gpurun_out/r3_bench15.err; tail -c 1800 gpurun_out/r3_bench15.json
timeout 900 python bench.py --dp-selftest --no-variants --no-cpu-baseline > gpurun_out/r3_dpself15.json 2> gpurun_out/r3_dpself15.err; python - <<'PY'
import json
try:
    d=json.loads(open("gpurun_out/r3_dpself15.json").read().strip().splitlines()[-1]); print("dpself", d["value"], d.get("data_parallel"), d["host_enqueue_ms_per_step"], d["host_cpu_ms_per_step"])
except Exception as e: print("dpself failed", e, open("gpurun_out/r3_dpself15.err").read()[-1500:])
PY
taskset -c 0-1 timeout 900 python bench.py --no-variants --no-cpu-baseline > gpurun_out/r3_bench15_2cores.json 2> /dev/null; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3_bench15_2cores.json").read().strip().splitlines()[-1]); print("2 cores", d["value"], d["host_enqueue_ms_per_step"], d["host_cpu_ms_per_step"], d["host_cores_available"])
PY
