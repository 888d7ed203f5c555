#!/bin/bash
# round 6: fp32 GEMM with the next tile's LDS-DMA pieces spread between the matrix instructions (libytvln_spread.so) against the shipped burst form:
# correctness (GEMM + model goldens), per-shape A/B under the fair protocol, headline ABAB
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
ALT=$PWD/youtube-vln_amd/ytvln/lib/libytvln_spread.so
YTVLN_LIB=$ALT timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -x -q -k "gemm or g0 or g2 or g11 or graph_replay" 2>&1 | tail -12
for rep in 1; do for v in base spread; do
if [ $v = base ]; then unset YTVLN_LIB; else export YTVLN_LIB=$ALT; fi
timeout 600 python bench.py --no-variants --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r6_spread_$v.json 2>/dev/null
python - <<PY
import json
d=json.loads(open('gpurun_out/r6_spread_$v.json').read().strip().splitlines()[-1]); f=d['roofline']['families']['gemm']; print('HEADLINE $v', d['value'], d['ms_per_step'], 'gemm ms', f['ms_per_step'], f['frac'])
PY
done; done
unset YTVLN_LIB
