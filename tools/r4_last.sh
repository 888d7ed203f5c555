#!/bin/bash
# last-call checks of the round: the GPU suite once more (a race shows up as flakiness), N = 2 functional record over gloo on one GPU, one-rank RCCL line
mkdir -p gpurun_out
python -m pytest tests -q -m gpu > gpurun_out/gpu_tests_rerun.log 2>&1; grep -E "passed|failed" gpurun_out/gpu_tests_rerun.log | tail -1
YTVLN_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 6 --warmup 2 --no-cpu-baseline --no-variants --no-kernel-timing --host-probe 0 > gpurun_out/round4_dp2_gloo_bench.json 2> gpurun_out/round4_dp2_gloo_bench.err
python bench.py --dp-selftest --steps 8 --warmup 3 --no-cpu-baseline --no-variants --host-probe 0 > gpurun_out/round4_dp_selftest_bench.json 2> gpurun_out/round4_dp_selftest_bench.err
python - <<P
import json
for f in ("round4_dp2_gloo_bench", "round4_dp_selftest_bench"):
    try:
        d = json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        print(f, d["n_gpus"], d["value"], d["ms_per_step"], d["config"]["execution"][:170])
    except Exception as e:
        print(f, "FAILED", e, open(f"gpurun_out/{f}.err").read()[-600:])
P
