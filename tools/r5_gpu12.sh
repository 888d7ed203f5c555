#!/bin/bash
# persistent form of the bf16 GEMM: tests, per-shape A/B against the launch-per-tile form and the vendor library, cfg5 / cfg2 bf16 step ABAB
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_bf16_gpu.py -m gpu -x -q -k "gemm or linear" 2>&1 | tail -3
for p in 0 1 0 1; do echo "== GEMM_BF16_PERSIST=$p"; YTVLN_GEMM_BF16_PERSIST=$p timeout 600 python tools/gemm_bf16_vs_torch.py 2>&1 | grep -v amdgpu.ids; done
for rep in 1 2; do for p in 0 1; do
YTVLN_GEMM_BF16_PERSIST=$p timeout 900 python bench.py --workload cfg5_long_traj_bs32 --precision bf16 --no-variants --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/r5k_cfg5_p$p.json 2>/dev/null
python - <<PY
import json
d=json.loads(open('gpurun_out/r5k_cfg5_p$p.json').read().strip().splitlines()[-1]); print('CFG5 bf16 persist=$p', d['value'], d['ms_per_step'], d['roofline']['frac'])
PY
done; done
for rep in 1 2; do for p in 0 1; do
YTVLN_GEMM_BF16_PERSIST=$p timeout 900 python bench.py --precision bf16 --no-variants --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r5k_cfg2_p$p.json 2>/dev/null
python - <<PY
import json
d=json.loads(open('gpurun_out/r5k_cfg2_p$p.json').read().strip().splitlines()[-1]); print('CFG2 bf16 persist=$p', d['value'], d['ms_per_step'], d['roofline']['frac'])
PY
done; done
