#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gemm_sk_gpu.py tests/test_bf16_gpu.py tests/test_kernels_gpu.py -q -m gpu -x -k "persistent or gemm or linear or ffn" > gpurun_out/r5d_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r5d_tests.log; tail -4 gpurun_out/r5d_tests.log
CONFIGS=old,old4,old0,sk0,dp0,sk0g1,sk3 PROBE=1 timeout 900 python tools/gemm_sk_bench.py > gpurun_out/r5d_sk_bench.log 2>&1
grep -v probe gpurun_out/r5d_sk_bench.log
grep "probe sk0:" gpurun_out/r5d_sk_bench.log | head -20
