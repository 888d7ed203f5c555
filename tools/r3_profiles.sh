#!/bin/bash
# round-3 evidence pass: default bench line (with variants + CPU baseline), kernel trace, PMC counters, other workloads
mkdir -p gpurun_out
timeout 1500 python bench.py > gpurun_out/round3_bench.json 2> gpurun_out/round3_bench.err; tail -c 600 gpurun_out/round3_bench.json
bash tools/kernel_stats.sh round3 > gpurun_out/round3_kernel_stats.txt 2>&1; head -30 gpurun_out/round3_kernel_stats.txt
PMC_TAG=r3_ bash tools/pmc_bench.sh > gpurun_out/round3_pmc_log.txt 2>&1
python tools/pmc_summary.py r3_ gpurun_out/round3_pmc_summary.json "PMC_TAG=r3_ tools/pmc_bench.sh (round-3 tree, default bench workload, eager launches, one --pmc pass per counter set)" | head -c 600
for W in cfg4_finetune_rank_bs16 infer_rerank_beam30 cfg1_tiny_mlm_bs2; do
  timeout 900 python bench.py --workload $W --no-variants --no-cpu-baseline > gpurun_out/round3_${W}_bench.json 2>/dev/null; cut -c1-160 gpurun_out/round3_${W}_bench.json
done
timeout 900 python bench.py --workload cfg5_long_traj_bs32 --precision bf16 --no-variants --no-cpu-baseline > gpurun_out/round3_cfg5_bf16_bench.json 2>/dev/null; cut -c1-160 gpurun_out/round3_cfg5_bf16_bench.json
timeout 900 python bench.py --precision fp32x3 --no-variants --no-cpu-baseline > gpurun_out/round3_fp32x3_bench.json 2>/dev/null; cut -c1-160 gpurun_out/round3_fp32x3_bench.json
