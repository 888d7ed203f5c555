#!/bin/bash
# cfg 5 (bf16-resident, 224 pairs x 576 regions) with the per-launch choice of GEMM form / store path against the round-5 forms, ABAB
export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_bf16_gpu.py tests/test_model_gpu.py -m gpu -x -q -k "bf16" 2>&1 | tail -2
for rep in 1 2; do for v in auto old; do
if [ $v = old ]; then export YTVLN_GEMM_BF16_FORM=0 YTVLN_GEMM_BF16_WIDE=0; else unset YTVLN_GEMM_BF16_FORM YTVLN_GEMM_BF16_WIDE; fi
timeout 900 python bench.py --workload cfg5_long_traj_bs32 --precision bf16 --no-variants --no-cpu-baseline --steps 6 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=d['roofline']['families']; print('CFG5 $v', d['value'], d['ms_per_step'], 'gemm', f['gemm']['ms_per_step'], f['gemm']['frac'], 'attn', f['attention']['ms_per_step'])"
done; done
