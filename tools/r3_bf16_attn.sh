#!/bin/bash
mkdir -p gpurun_out
{
for l in scratch/lib_oldattn.so "" scratch/lib_oldattn.so; do
  echo "== YTVLN_LIB=$l PRECISION=bf16 cfg5 shapes"
  YTVLN_LIB=$l PRECISION=bf16 PAIRS_N=224 REGIONS=576 timeout 600 python tools/attn_bench.py 2>&1 | grep -v "^\[\|amdgpu.ids"
done
echo "== d=64 one-wave kernels: forms + timing"
YTVLN_ATTN_DSPLIT=0 KNOB=YTVLN_ATTN_W1_D64 A=0 B=1 BWD=1 YTVLN_ATTN_W1_DKV=2 timeout 600 python tools/attn_form_check.py 2>&1 | grep -v amdgpu.ids | grep -v "close "
for v in 0 1 1 0; do
  echo "-- YTVLN_ATTN_W1_D64=$v"; YTVLN_ATTN_W1_D64=$v CASES=txt timeout 300 python tools/attn_bench.py 2>&1 | grep "txt self"
done
timeout 1200 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k attention 2>&1 | grep "passed\|failed"
} > gpurun_out/bf16_attn3.log 2>&1
cat gpurun_out/bf16_attn3.log
