#!/bin/bash
# round-2 GPU pass 17: the attention / bias-gradient knobs' alternate paths through the kernel and model parity tests
mkdir -p gpurun_out
for V in "YTVLN_ATTN_WAVES=4" "YTVLN_ATTN_WAVES=3" "YTVLN_ATTN_WAVES=1" "YTVLN_ATTN_PAIRS=2 YTVLN_ATTN_DKV_STAGES=2" "YTVLN_ATTN_PAIRS=2 YTVLN_ATTN_DKV_STAGES=1" "YTVLN_ATTN_PAIRS=1 YTVLN_ATTN_DKV_STAGES=2" "YTVLN_ATTN_DSPLIT=0" "YTVLN_ATTN_DELTA_KERNEL=1" "YTVLN_FUSED_BIAS_GRAD=0"; do
  echo "== $V"; env $V timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "attention or g0 or g2_full_model_all or g4" 2>&1 | tail -1
done > gpurun_out/r2_knob_matrix.log 2>&1
cat gpurun_out/r2_knob_matrix.log
