#!/bin/bash
# step-level ABBA of the one-wave-per-SIMD attention kernels (d = 64 text stream on / off; everything off) + the GPU tests with the defaults
mkdir -p gpurun_out
{
for v in "1 1" "1 0" "0 0" "1 1"; do
  set -- $v
  echo "== one-wave kernels $1 (YTVLN_ATTN_W1 = _DQ = _DKV), YTVLN_ATTN_W1_D64=$2"
  YTVLN_ATTN_W1=$1 YTVLN_ATTN_W1_DQ=$1 YTVLN_ATTN_W1_DKV=$1 YTVLN_ATTN_W1_D64=$2 timeout 900 python bench.py --steps 10 --warmup 3 --no-variants 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['families']['attention']['ms_per_step'], d['roofline']['families']['attention']['frac'])"
done
timeout 2700 python -m pytest tests/ -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1; grep "passed\|failed" gpurun_out/gpu_tests.log | tail -2
} > gpurun_out/w1_step.log 2>&1
cat gpurun_out/w1_step.log
