#!/bin/bash
# cfg 4 (fine-tune, 24192 image rows: 380 tiles of 256x256 = 1.48 rounds): the persistent / stream-K GEMM options against the default plans, ABAB
export TMPDIR=/tmp; mkdir -p gpurun_out
for rep in 1 2; do for v in "default" "YTVLN_GEMM_SK=1" "YTVLN_GEMM_SK=3 YTVLN_GEMM_SK_TILE=4"; do
env $( [ "$v" = default ] || echo $v ) timeout 600 python bench.py --workload cfg4_finetune_rank_bs16 --no-variants --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=d['roofline']['families']; print('CFG4 [$v]', d['value'], d['ms_per_step'], 'gemm', f['gemm']['ms_per_step'], f['gemm']['frac'])"
done; done 2>&1 | tee gpurun_out/r6_cfg4_sk_ab.log
