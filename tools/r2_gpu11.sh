#!/bin/bash
# round-2 GPU pass 11: full suite on the current build, cfg5 bf16 with the two-stage dK/dV variant, refreshed headline artefacts
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/round2_gpu_tests.log 2>&1; echo "rc=$?" >> gpurun_out/round2_gpu_tests.log
tail -3 gpurun_out/round2_gpu_tests.log
timeout 900 python bench.py > gpurun_out/round2_bench.json 2> gpurun_out/round2_bench.err; cut -c1-240 gpurun_out/round2_bench.json
TOPN=45 bash tools/kernel_stats.sh round2 > gpurun_out/round2_kernel_stats.txt 2>&1; head -14 gpurun_out/round2_kernel_stats.txt | cut -c1-150
bash tools/attn_pmc.sh round2_img img > gpurun_out/round2_img_attn_pmc.txt 2>&1; tail -4 gpurun_out/round2_img_attn_pmc.txt | cut -c1-200
bash tools/attn_pmc.sh round2_co co > gpurun_out/round2_co_attn_pmc.txt 2>&1; tail -4 gpurun_out/round2_co_attn_pmc.txt | cut -c1-200
timeout 900 python bench.py --workload cfg5_long_traj_bs32 --precision bf16 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/round2_cfg5_bf16_bench.json 2> gpurun_out/round2_cfg5.err; cut -c1-160 gpurun_out/round2_cfg5_bf16_bench.json
TOPN=14 bash tools/kernel_stats.sh round2_cfg5_bf16 --workload cfg5_long_traj_bs32 --precision bf16 > gpurun_out/round2_cfg5_bf16_kernel_stats.txt 2>&1; head -12 gpurun_out/round2_cfg5_bf16_kernel_stats.txt | cut -c1-150
{ echo "== default"; timeout 300 python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids; } > gpurun_out/round2_attn_bench.log 2>&1; cat gpurun_out/round2_attn_bench.log
