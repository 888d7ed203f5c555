#!/bin/bash
# the 224x256 fp32 tile (GEMM_T224=1: planner may take it): tests, text shapes under the fair protocol, headline ABAB
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gemm_sk_gpu.py tests/test_kernels_gpu.py -m gpu -x -q -k "224 or gemm or linear or ffn" 2>&1 | tail -3
for v in 0 1 0 1; do echo "== GEMM_T224=$v"; YTVLN_GEMM_T224=$v SHAPES=text CONFIGS=old timeout 600 python tools/gemm_sk_bench.py 2>&1 | grep -v amdgpu.ids | tail -14; done
for rep in 1 2 3; do for v in 0 1; do
YTVLN_GEMM_T224=$v timeout 600 python bench.py --no-variants --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r5p_bench_t$v.json 2>/dev/null
python - <<PY
import json
d=json.loads(open('gpurun_out/r5p_bench_t$v.json').read().strip().splitlines()[-1]); print('HEADLINE GEMM_T224=$v', d['value'], d['ms_per_step'], d['roofline']['frac'])
PY
done; done
