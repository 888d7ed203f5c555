#!/bin/bash
mkdir -p gpurun_out
{
BWD=1 KNOB=YTVLN_ATTN_W1_DQ timeout 600 python tools/attn_form_check.py 2>&1 | grep -v amdgpu.ids | grep -v "close "
YTVLN_ATTN_W1_DQ=1 timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "attention" 2>&1 | tail -8
for cfg in "0 0" "0 1" "0 1" "0 0"; do
  set -- $cfg
  echo "== YTVLN_ATTN_W1=$1 YTVLN_ATTN_W1_DQ=$2"
  YTVLN_ATTN_W1=$1 YTVLN_ATTN_W1_DQ=$2 timeout 300 python tools/attn_bench.py 2>&1 | grep -v "^\[\|amdgpu.ids"
done
YTVLN_LIB=scratch/lib_w1t.so YTVLN_ATTN_W1=0 YTVLN_ATTN_W1_DQ=1 CASES=img timeout 300 python tools/attn_bench.py 2>&1 | grep "w1 dq timing" | tail -3
} > gpurun_out/attn_dq.log 2>&1
tail -44 gpurun_out/attn_dq.log
