#!/bin/bash
# Forward / dQ attention kernels at 1..4 waves per workgroup (YTVLN_ATTN_WAVES), ABBA around the default.
mkdir -p gpurun_out
{
for w in 0 1 2 3 4 1 0; do
  echo "== YTVLN_ATTN_WAVES=$w"
  YTVLN_ATTN_WAVES=$w CASES=img timeout 300 python tools/attn_bench.py 2>&1 | grep -v "^\[" | head -3
done
for w in 0 1 0; do
  echo "== co, YTVLN_ATTN_WAVES=$w"
  YTVLN_ATTN_WAVES=$w CASES=co timeout 300 python tools/attn_bench.py 2>&1 | grep -v "^\[" | head -6
done
} > gpurun_out/attn_waves.log 2>&1
tail -40 gpurun_out/attn_waves.log
