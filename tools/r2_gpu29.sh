#!/bin/bash
# round-2 GPU pass 29: phased mode records -- full-size one-rank self-test (default = phased), 2 gloo ranks sharing the GPU in phased mode
mkdir -p gpurun_out
timeout 900 python bench.py --dp-selftest --steps 10 --warmup 3 > gpurun_out/round2_dp_selftest_bench.json 2>/dev/null; cut -c1-200 gpurun_out/round2_dp_selftest_bench.json
YTVLN_DIST_BACKEND=gloo YTVLN_DP_GRAPH=phased timeout 1200 python bench.py --gpus 2 --steps 4 --warmup 1 --no-cpu-baseline --no-variants > gpurun_out/round2_dp2_gloo_phased_bench.json 2> gpurun_out/round2_dp2_gloo_phased.err; cut -c1-200 gpurun_out/round2_dp2_gloo_phased_bench.json; tail -3 gpurun_out/round2_dp2_gloo_phased.err | cut -c1-300
