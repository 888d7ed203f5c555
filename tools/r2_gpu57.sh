#!/bin/bash
# round-2 GPU pass 57: the round-2 planner / layout knobs at their "old behaviour" values through the GEMM tests and the model goldens
mkdir -p gpurun_out
for V in "YTVLN_GEMM_BIG_TA=0" "YTVLN_GEMM_SPLIT_MAP=0" "YTVLN_BF16_BIG_SPLIT=0" "YTVLN_GEMM_BIG_TA=0 YTVLN_GEMM_SPLIT_MAP=0 YTVLN_FUSED_BIAS_GRAD=0"; do
  echo "== $V"; env $V timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "gemm or g0 or g2_full_model_all or g4 or bf16 or g11" 2>&1 | grep "passed\|failed" | tail -1
done > gpurun_out/r2_knob_matrix2.log 2>&1
cat gpurun_out/r2_knob_matrix2.log
