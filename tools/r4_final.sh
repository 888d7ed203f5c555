#!/bin/bash
# final-tree evidence: full GPU test log, default bench line, N = 8 host rehearsal
set -u
mkdir -p gpurun_out
python -m pytest tests -q -m gpu > gpurun_out/round4_gpu_tests.log 2>&1; grep -E "passed|failed" gpurun_out/round4_gpu_tests.log | tail -2
timeout 1200 python bench.py > gpurun_out/round4c_bench_default.json 2> gpurun_out/round4c_bench_default.err
python - <<P
import json
d = json.loads(open("gpurun_out/round4c_bench_default.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["power"], {k: v["value"] for k, v in d["variants"].items()})
P
timeout 1500 python tools/host8.py 8 2 12 > gpurun_out/host8.log 2>&1; tail -3 gpurun_out/host8.log
