#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_model_gpu.py -q -m gpu -k "g15 or g14" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_rccl_gpu.py -q -m gpu 2>&1 | tail -2
timeout 900 python bench.py --dp-selftest --no-variants --no-cpu-baseline > gpurun_out/r3_dpself15.json 2> gpurun_out/r3_dpself15.err; python - <<'PY'
import json
L=open("gpurun_out/r3_dpself15.json").read().strip().splitlines(); print(len(L),"stdout lines")
d=json.loads(L[-1]); print("dpself", d["value"], json.dumps(d.get("data_parallel")), d["host_enqueue_ms_per_step"], d["host_cpu_ms_per_step"])
PY
grep -h "Bytes -> Algo\|coll channels\|via " /tmp/ytvln_rccl_*.log 2>/dev/null | head -8 | cut -c1-220
