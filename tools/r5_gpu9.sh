#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
YTVLN_GEMM_SW=1 timeout 1500 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "gemm or linear or ffn" > gpurun_out/r5h_tests.log 2>&1
echo "tests (GEMM_SW=1) rc=$?" >> gpurun_out/r5h_tests.log; tail -3 gpurun_out/r5h_tests.log
SHAPES=all3 CONFIGS=old,sw4,old4,sw3,old3 timeout 1200 python tools/gemm_sk_bench.py > gpurun_out/r5h_sw_bench.log 2>&1
cat gpurun_out/r5h_sw_bench.log
for sw in 0 1 2 0 1; do
YTVLN_GEMM_SW=$sw timeout 600 python bench.py --no-variants --no-cpu-baseline --steps 16 --warmup 4 > gpurun_out/r5h_bench_sw$sw.json 2>/dev/null
python - <<PY
import json
d=json.loads(open('gpurun_out/r5h_bench_sw$sw.json').read().strip().splitlines()[-1]); print('HEADLINE GEMM_SW=$sw', d['value'], d['ms_per_step'], d['roofline']['frac'])
PY
done
