#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
for sw in 0 3 0 3 0 3; do
YTVLN_GEMM_SW=$sw timeout 600 python bench.py --no-variants --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r5i_bench_sw$sw.json 2>/dev/null
python - <<PY
import json
d=json.loads(open('gpurun_out/r5i_bench_sw$sw.json').read().strip().splitlines()[-1]); print('HEADLINE GEMM_SW=$sw', d['value'], d['ms_per_step'], d['roofline']['frac'])
PY
done
