#!/bin/bash
# round-2 GPU pass 46: headline line re-run against the final counter summary; cfg5 bf16 line + trace on the final tree
mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/round2_bench.json 2> gpurun_out/round2_bench.err; cut -c1-200 gpurun_out/round2_bench.json
timeout 900 python bench.py --workload cfg5_long_traj_bs32 --precision bf16 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/round2_cfg5_bf16_bench.json 2> gpurun_out/round2_cfg5.err; cut -c1-160 gpurun_out/round2_cfg5_bf16_bench.json
TOPN=12 bash tools/kernel_stats.sh round2_cfg5_bf16 --workload cfg5_long_traj_bs32 --precision bf16 > gpurun_out/round2_cfg5_bf16_kernel_stats.txt 2>&1; head -12 gpurun_out/round2_cfg5_bf16_kernel_stats.txt | cut -c1-150
