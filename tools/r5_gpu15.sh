#!/bin/bash
# fp32 256x256 GEMM: ping-pong main loop (GEMM_PP=1) against the lock-step loop -- tests, per-shape fair protocol, headline ABAB
mkdir -p gpurun_out; export TMPDIR=/tmp
YTVLN_GEMM_PP=1 timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_gemm_sk_gpu.py -m gpu -x -q -k "gemm or linear or ffn or persistent" 2>&1 | tail -3
for v in 0 1 0 1; do echo "== GEMM_PP=$v"; YTVLN_GEMM_PP=$v SHAPES=all3 CONFIGS=old timeout 600 python tools/gemm_sk_bench.py 2>&1 | grep -v amdgpu.ids | tail -22; done
for rep in 1 2 3; do for v in 0 1; do
YTVLN_GEMM_PP=$v timeout 600 python bench.py --no-variants --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r5l_bench_pp$v.json 2>/dev/null
python - <<PY
import json
d=json.loads(open('gpurun_out/r5l_bench_pp$v.json').read().strip().splitlines()[-1]); print('HEADLINE GEMM_PP=$v', d['value'], d['ms_per_step'], d['roofline']['frac'])
PY
done; done
