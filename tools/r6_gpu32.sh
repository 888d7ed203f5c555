#!/bin/bash
# the opt-in run-time options against the full-model goldens (each setting: g2 fp32 / g10 fp32+bf16 / g16b bf16)
export TMPDIR=/tmp; mkdir -p gpurun_out
run() { echo "== $*"; env "$@" timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -x -q -k "g2_full_model_all_losses or g10 or (g16b and bf16)" 2>&1 | grep -a "passed\|failed\|Error" | tail -2; }
{
run YTVLN_GEMM_SK=1
run YTVLN_GEMM_SK=3 YTVLN_GEMM_SK_TILE=4
run YTVLN_GEMM_SW=1
run YTVLN_ATTN_DKV_SPLIT=1
run YTVLN_GEMM_BF16_FORM=4
run YTVLN_GEMM_BF16_FORM=-1 YTVLN_GEMM_BF16_WIDE=-1
run YTVLN_GEMM_BF16_FORM=3
run YTVLN_GEMM_BF16_FORM=2 YTVLN_GEMM_BF16_WIDE=1
run YTVLN_GEMM_STAGGER=20
} > gpurun_out/r6_option_sweep.log 2>&1
cat gpurun_out/r6_option_sweep.log
