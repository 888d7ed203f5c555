#!/bin/bash
# the fragment-load form (load_rowfrag) on the bf16 attention kernels at the cfg-5 shapes, and the one-wave form tests
mkdir -p gpurun_out
{
for l in "" scratch/lib_branchy.so; do
  echo "== YTVLN_LIB=$l PRECISION=bf16 cfg5 shapes"
  YTVLN_LIB=$l PRECISION=bf16 PAIRS_N=224 REGIONS=576 timeout 600 python tools/attn_bench.py 2>&1 | grep -v "^\[\|amdgpu.ids"
done
timeout 1200 python -m pytest tests/test_attention_forms_gpu.py -x -q -m gpu 2>&1 | grep "passed\|failed"
timeout 1200 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k attention 2>&1 | grep "passed\|failed"
} > gpurun_out/bf16_attn2.log 2>&1
cat gpurun_out/bf16_attn2.log
