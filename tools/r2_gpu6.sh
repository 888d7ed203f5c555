#!/bin/bash
# round-2 GPU pass 6: full-size config tests (fp32 / bf16 / fp32x3), attention counters, cfg5 bf16 + cfg4 bench lines
mkdir -p gpurun_out
timeout 3000 python -m pytest tests/test_model_gpu.py tests/test_kernels_gpu.py -x -q -m gpu -k "g10 or g11 or g12 or fp32x3 or rowsum" > gpurun_out/r2_cfg_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r2_cfg_tests.log
tail -12 gpurun_out/r2_cfg_tests.log
bash tools/attn_pmc.sh r2c img > gpurun_out/r2c_attn_pmc.txt 2>&1; tail -8 gpurun_out/r2c_attn_pmc.txt
bash tools/attn_pmc.sh r2c_co co > gpurun_out/r2c_co_attn_pmc.txt 2>&1; tail -8 gpurun_out/r2c_co_attn_pmc.txt
timeout 900 python bench.py --workload cfg5_long_traj_bs32 --precision bf16 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r2_cfg5_bf16.json 2> gpurun_out/r2_cfg5_bf16.err; cut -c1-200 gpurun_out/r2_cfg5_bf16.json; tail -2 gpurun_out/r2_cfg5_bf16.err
timeout 900 python bench.py --workload cfg4_finetune_rank_bs16 --steps 8 --warmup 3 --no-cpu-baseline --no-variants > gpurun_out/r2_cfg4.json 2> gpurun_out/r2_cfg4.err; cut -c1-200 gpurun_out/r2_cfg4.json
