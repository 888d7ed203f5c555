#!/bin/bash
# long-K plans: GEMM tests on the new library, then the headline against the previous library (libytvln_prevplan.so), ABAB
export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_gemm_sk_gpu.py tests/test_abi.py -m gpu -x -q 2>&1 | grep -a "passed\|failed" | tail -2
bash tools/r6_gpu11.sh base prevplan | tee gpurun_out/r6_long_k_headline_ab.log
