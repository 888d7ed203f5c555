#!/bin/bash
mkdir -p gpurun_out
{
YTVLN_ATTN_DSPLIT=0 KNOB=YTVLN_ATTN_W1 A=0 B=2 timeout 600 python tools/attn_form_check.py 2>&1 | grep -v amdgpu.ids | grep -v "close "
for v in 0 2 1 2 0; do
  echo "== YTVLN_ATTN_W1=$v"
  YTVLN_ATTN_W1=$v FWD_ONLY=1 timeout 300 python tools/attn_bench.py 2>&1 | grep -v "^\[\|amdgpu.ids" | grep "img\|pair\|co "
done
} > gpurun_out/fwd.log 2>&1
cat gpurun_out/fwd.log
