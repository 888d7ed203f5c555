#!/bin/bash
# after returning the bf16 GEMM defaults to form 0 / 2-byte stores: the bf16 tests, cfg 5 and the headline once each
export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_bf16_gpu.py tests/test_model_gpu.py tests/test_gemm_sk_gpu.py -m gpu -x -q 2>&1 | tail -2
for w in "cfg5_long_traj_bs32 --precision bf16" "cfg2_full_pretrain_bs8"; do
timeout 900 python bench.py --workload $w --no-variants --no-cpu-baseline --steps 6 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=d['roofline']['families']; print('BENCH', d['config']['workload'], d['value'], d['ms_per_step'], 'gemm', f['gemm']['ms_per_step'], f['gemm']['frac'], 'attn', f['attention']['ms_per_step'])"
done
