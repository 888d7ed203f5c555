#!/bin/bash
# round-2 GPU pass 41: bf16 weight gradients on 256x256 split-K tiles -- bf16 GEMM tests, cfg5 A/B
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "bf16" 2>&1 | tail -2
timeout 1500 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "bf16" 2>&1 | grep "passed\|failed" | tail -2
for rep in 1 2; do for v in 0 1; do YTVLN_BF16_BIG_SPLIT=$v timeout 900 python bench.py --workload cfg5_long_traj_bs32 --precision bf16 --steps 6 --warmup 2 --no-cpu-baseline --no-variants --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][0]); print('big_split=$v', d['value'], d['ms_per_step'], d['final_loss'])"; done; done
