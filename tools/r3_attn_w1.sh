#!/bin/bash
mkdir -p gpurun_out
{
YTVLN_ATTN_DSPLIT=0 timeout 600 python tools/attn_form_check.py 2>&1 | grep -v amdgpu.ids
for w in 0 1 1 0; do
  echo "== YTVLN_ATTN_W1=$w"
  YTVLN_ATTN_W1=$w FWD_ONLY=1 timeout 300 python tools/attn_bench.py 2>&1 | grep -v "^\[\|amdgpu.ids"
done
if [ -f scratch/lib_w1t.so ]; then
  echo "== timing build"
  YTVLN_LIB=scratch/lib_w1t.so YTVLN_ATTN_W1=1 FWD_ONLY=1 CASES=img timeout 300 python tools/attn_bench.py 2>&1 | grep "w1 timing" | tail -4
fi
} > gpurun_out/attn_w1.log 2>&1
tail -40 gpurun_out/attn_w1.log
