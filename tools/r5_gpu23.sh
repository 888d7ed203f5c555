#!/bin/bash
# split-K on the 160-/224-row tiles (N = 768 text shapes): tests, text shapes, headline ABAB against the previous build
mkdir -p gpurun_out; export TMPDIR=/tmp
BASE=$PWD/youtube-vln_amd/ytvln/lib/libytvln_base.so
timeout 1200 python -m pytest tests/test_gemm_sk_gpu.py tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -x -q -k "224 or gemm or linear or ffn or g0 or g2 or g11 or replay" 2>&1 | tail -3
for v in base new base new; do
if [ $v = base ]; then export YTVLN_LIB=$BASE; else unset YTVLN_LIB; fi
echo "== $v"; SHAPES=textco CONFIGS=old timeout 600 python tools/gemm_sk_bench.py 2>&1 | grep -v amdgpu.ids | tail -17
done
for rep in 1 2 3; do for v in base new; do
if [ $v = base ]; then export YTVLN_LIB=$BASE; else unset YTVLN_LIB; fi
timeout 600 python bench.py --no-variants --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r5q_bench_$v.json 2>/dev/null
python - <<PY
import json
d=json.loads(open('gpurun_out/r5q_bench_$v.json').read().strip().splitlines()[-1]); print('HEADLINE $v', d['value'], d['ms_per_step'], d['roofline']['frac'])
PY
done; done
