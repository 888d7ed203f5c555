#!/bin/bash
# bf16 LayerNorm kernels with next-row prefetch against the committed ones: tests, cfg 5 / cfg 2 bf16 step ABAB
mkdir -p gpurun_out; export TMPDIR=/tmp
BASE=$PWD/youtube-vln_amd/ytvln/lib/libytvln_base.so
timeout 1200 python -m pytest tests/test_bf16_gpu.py tests/test_model_gpu.py -m gpu -x -q -k "attention or attn or bf16 or g10 or g16" 2>&1 | tail -3
for rep in 1 2 3; do for v in base new; do
if [ $v = base ]; then export YTVLN_LIB=$BASE; else unset YTVLN_LIB; fi
timeout 900 python bench.py --workload cfg5_long_traj_bs32 --precision bf16 --no-variants --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/r5n_cfg5_$v.json 2>/dev/null
python - <<PY
import json
d=json.loads(open('gpurun_out/r5n_cfg5_$v.json').read().strip().splitlines()[-1]); f=d['roofline'].get('families',{}).get('layernorm',{}); print('CFG5 bf16 $v', d['value'], d['ms_per_step'], 'LN ms', f.get('ms_per_step'), f.get('frac'))
PY
done; done
for rep in 1 2; do for v in base new; do
if [ $v = base ]; then export YTVLN_LIB=$BASE; else unset YTVLN_LIB; fi
timeout 900 python bench.py --precision bf16 --no-variants --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r5n_cfg2_$v.json 2>/dev/null
python - <<PY
import json
d=json.loads(open('gpurun_out/r5n_cfg2_$v.json').read().strip().splitlines()[-1]); f=d['roofline'].get('families',{}).get('layernorm',{}); print('CFG2 bf16 $v', d['value'], d['ms_per_step'], 'LN ms', f.get('ms_per_step'), f.get('frac'))
PY
done; done
