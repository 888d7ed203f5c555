#!/bin/bash
# co-attention pair launches with the long workgroups first (dK/dV: the long-query direction): tests, headline ABAB
mkdir -p gpurun_out; export TMPDIR=/tmp
BASE=$PWD/youtube-vln_amd/ytvln/lib/libytvln_base.so
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_attention_forms_gpu.py tests/test_model_gpu.py -m gpu -x -q -k "attention or attn or g0 or g2 or g11" 2>&1 | tail -3
for rep in 1 2 3; do for v in base new; do
if [ $v = base ]; then export YTVLN_LIB=$BASE; else unset YTVLN_LIB; fi
timeout 600 python bench.py --no-variants --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r5r_bench_$v.json 2>/dev/null
python - <<PY
import json
d=json.loads(open('gpurun_out/r5r_bench_$v.json').read().strip().splitlines()[-1]); f=d['roofline']['families']['attention']; print('HEADLINE $v', d['value'], d['ms_per_step'], 'attention ms', f['ms_per_step'], f['frac'])
PY
done; done
