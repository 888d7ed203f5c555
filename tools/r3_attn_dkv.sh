#!/bin/bash
mkdir -p gpurun_out
{
BWD=1 KNOB=YTVLN_ATTN_W1_DKV timeout 600 python tools/attn_form_check.py 2>&1 | grep -v amdgpu.ids | grep -v "close "
YTVLN_ATTN_W1_DKV=1 timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "attention" 2>&1 | grep "passed\|failed"
for v in 0 1 1 0; do
  echo "== YTVLN_ATTN_W1_DKV=$v"
  YTVLN_ATTN_W1_DKV=$v timeout 300 python tools/attn_bench.py 2>&1 | grep -v "^\[\|amdgpu.ids"
done
} > gpurun_out/attn_dkv.log 2>&1
tail -44 gpurun_out/attn_dkv.log
