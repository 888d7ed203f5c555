#!/bin/bash
# round 6, call 4: bf16 GEMM form 4 (four waves of 128x128, one per SIMD): correctness on the full grids, A/B + K sweep vs form 0 and the library
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
YTVLN_GEMM_BF16_FORM=4 timeout 900 python -m pytest tests/test_bf16_gpu.py -m gpu -x -q -k "gemm" 2>&1 | tail -3
CONFIGS=${CONFIGS:-0:0,4:0} KSWEEP=1 SHAPES="${SHAPES:-fwd,dX,decoder}" timeout 1200 python tools/gemm_bf16_forms.py > gpurun_out/r6_gemm_bf16_form4.log 2>&1; cat gpurun_out/r6_gemm_bf16_form4.log
