#!/bin/bash
# step-level ABBA of the one-wave-per-SIMD dQ kernel + the GPU tests with it on (the default)
mkdir -p gpurun_out
{
for v in 1 0 0 1; do
  echo "== YTVLN_ATTN_W1_DQ=$v"
  YTVLN_ATTN_W1_DQ=$v timeout 900 python bench.py --steps 10 --warmup 3 --no-variants 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['families']['attention']['ms_per_step'], d['roofline']['families']['attention']['frac'])"
done
timeout 2400 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -4
} > gpurun_out/w1_step.log 2>&1
tail -30 gpurun_out/w1_step.log
