#!/bin/bash
# round 5, GPU call 2: spill-free epilogue (all fp32 GEMM kernels) + persistent kernel v2
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gemm_sk_gpu.py tests/test_kernels_gpu.py -x -q -m gpu -k "gemm or linear or ffn or persistent" > gpurun_out/r5b_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r5b_tests.log
tail -5 gpurun_out/r5b_tests.log
PROBE=1 timeout 600 python tools/gemm_sk_bench.py > gpurun_out/r5b_sk_bench.log 2>&1
echo "bench rc=$?" >> gpurun_out/r5b_sk_bench.log
grep -v probe gpurun_out/r5b_sk_bench.log
YTVLN_GEMM_SK=0 timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/r5b_bench_sk0.json 2> gpurun_out/r5b_bench_sk0.err
echo "bench sk0 rc=$?"; python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r5b_bench_sk0.json').read().strip().splitlines()[-1]); print('SK0', d['value'], d['ms_per_step'], d['roofline']['frac'], {k:v.get('ms') for k,v in d['roofline'].get('families',{}).items()})
except Exception as e: print('parse fail', e)
PY
YTVLN_GEMM_SK=2 timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/r5b_bench_sk2.json 2> gpurun_out/r5b_bench_sk2.err
echo "bench sk2 rc=$?"; python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r5b_bench_sk2.json').read().strip().splitlines()[-1]); print('SK2', d['value'], d['ms_per_step'], d['roofline']['frac'], {k:v.get('ms') for k,v in d['roofline'].get('families',{}).items()})
except Exception as e: print('parse fail', e)
PY
