#!/bin/bash
mkdir -p gpurun_out
{
for i in 1 2; do
  timeout 900 python bench.py --steps 10 --warmup 3 --no-variants 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['families']['attention']['ms_per_step'], d['roofline']['families']['attention']['frac'])"
done
timeout 2700 python -m pytest tests/ -q -m gpu > gpurun_out/gpu_tests.log 2>&1; grep "passed\|failed" gpurun_out/gpu_tests.log | tail -2
for v in 0 1; do YTVLN_ATTN_W1_D64=$v CASES=txt timeout 300 python tools/attn_bench.py 2>&1 | grep "txt self"; done
} > gpurun_out/hash.log 2>&1
cat gpurun_out/hash.log
