#!/bin/bash
# round 5, GPU call 3: fair (warm, interleaved) per-shape comparison + all new tests + kernel-trace durations + headline
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gemm_sk_gpu.py tests/test_kernels_gpu.py tests/test_bf16_gpu.py -q -m gpu -k "gemm or linear or ffn or persistent or embedding_dropout or full_grids or layernorm" > gpurun_out/r5c_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r5c_tests.log
tail -8 gpurun_out/r5c_tests.log
timeout 900 python tools/gemm_sk_bench.py > gpurun_out/r5c_sk_bench.log 2>&1
echo "bench rc=$?" >> gpurun_out/r5c_sk_bench.log
cat gpurun_out/r5c_sk_bench.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r5c_bench.json 2> gpurun_out/r5c_bench.err
echo "bench rc=$?"; python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r5c_bench.json').read().strip().splitlines()[-1]); print('HEADLINE', d['value'], d['ms_per_step'], d['roofline']['frac'], {k:(v.get('ms_per_step'), v.get('frac')) for k,v in d['roofline'].get('families',{}).items()})
except Exception as e: print('parse fail', e)
PY
