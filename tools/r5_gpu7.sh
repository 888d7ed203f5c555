#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_bf16_gpu.py tests/test_bf16_host_gpu.py -q -m gpu > gpurun_out/r5f_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r5f_tests.log; tail -4 gpurun_out/r5f_tests.log
timeout 900 python -m pytest tests/test_model_gpu.py -q -m gpu -k "bf16 or g10 or g16" >> gpurun_out/r5f_tests.log 2>&1
echo "model tests rc=$?" >> gpurun_out/r5f_tests.log; tail -3 gpurun_out/r5f_tests.log
PRECISION=bf16 PAIRS_N=224 REGIONS=576 timeout 600 python tools/attn_bench.py > gpurun_out/r5f_attn_bf16_cfg5.log 2>&1; cat gpurun_out/r5f_attn_bf16_cfg5.log
timeout 900 python bench.py --workload cfg5_long_traj_bs32 --precision bf16 --no-variants --no-cpu-baseline --steps 6 --warmup 2 > gpurun_out/r5f_cfg5.json 2> gpurun_out/r5f_cfg5.err
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r5f_cfg5.json').read().strip().splitlines()[-1]); print('CFG5', d['value'], d['ms_per_step'], {k:(v.get('ms_per_step'), v.get('frac')) for k,v in d['roofline'].get('families',{}).items()})
except Exception as e: print('parse fail', e)
PY
