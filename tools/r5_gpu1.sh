#!/bin/bash
# round 5, GPU call 1: persistent GEMM correctness + per-shape A/B + in-kernel timeline
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gemm_sk_gpu.py -x -q -m gpu > gpurun_out/r5_sk_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r5_sk_tests.log
tail -5 gpurun_out/r5_sk_tests.log
PROBE=1 timeout 600 python tools/gemm_sk_bench.py > gpurun_out/r5_sk_bench.log 2>&1
echo "bench rc=$?" >> gpurun_out/r5_sk_bench.log
cat gpurun_out/r5_sk_bench.log
