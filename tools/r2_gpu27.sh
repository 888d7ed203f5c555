#!/bin/bash
# round-2 GPU pass 27: phased backward (exchange overlapped with the rest of backward) -- one-rank RCCL world tests, self-test timings
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_rccl_gpu.py -x -q -m gpu 2>&1 | tail -15
for M in split phased; do
  YTVLN_DP_GRAPH=$M timeout 900 python bench.py --dp-selftest --steps 10 --warmup 3 --no-cpu-baseline --no-variants > gpurun_out/r2_dpself_$M.json 2> gpurun_out/r2_dpself_$M.err
  python - <<PY
import json
try:
    d = json.loads([l for l in open("gpurun_out/r2_dpself_$M.json") if l.startswith("{")][0]); print("$M", d["value"], d["ms_per_step"], d["final_loss"], d["config"]["execution"])
except Exception as e:
    print("$M failed", e, open("gpurun_out/r2_dpself_$M.err").read()[-1500:])
PY
done
