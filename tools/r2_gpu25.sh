#!/bin/bash
# round-2 GPU pass 25: the whole suite + smoke on the current tree, then every round-2 artefact refreshed (bench lines, kernel traces, counters)
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/round2_gpu_tests.log 2>&1; echo "rc=$?" >> gpurun_out/round2_gpu_tests.log
tail -4 gpurun_out/round2_gpu_tests.log
python __graft_entry__.py smoke 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/round2_bench.json 2> gpurun_out/round2_bench.err; cut -c1-240 gpurun_out/round2_bench.json
TOPN=45 bash tools/kernel_stats.sh round2 > gpurun_out/round2_kernel_stats.txt 2>&1; head -16 gpurun_out/round2_kernel_stats.txt | cut -c1-150
PMC_TAG=r2_ bash tools/pmc_bench.sh > gpurun_out/round2_pmc_log.txt 2>&1
python tools/pmc_summary.py r2_ gpurun_out/round2_pmc_summary.json "rocprofv3 --pmc <counter set> --kernel-trace, one pass per counter set, over python bench.py --graph off --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-variants (2 training steps); MI355X; round-2 build (split-major split-K layout)"
timeout 900 python bench.py --workload cfg5_long_traj_bs32 --precision bf16 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/round2_cfg5_bf16_bench.json 2> gpurun_out/round2_cfg5.err; cut -c1-160 gpurun_out/round2_cfg5_bf16_bench.json
TOPN=25 bash tools/kernel_stats.sh round2_cfg5_bf16 --workload cfg5_long_traj_bs32 --precision bf16 > gpurun_out/round2_cfg5_bf16_kernel_stats.txt 2>&1; head -12 gpurun_out/round2_cfg5_bf16_kernel_stats.txt | cut -c1-150
timeout 900 python bench.py --workload cfg4_finetune_rank_bs16 --no-cpu-baseline --no-variants > gpurun_out/round2_cfg4_bench.json 2>/dev/null; cut -c1-160 gpurun_out/round2_cfg4_bench.json
timeout 900 python bench.py --precision fp32x3 --no-cpu-baseline > gpurun_out/round2_fp32x3_bench.json 2>/dev/null; cut -c1-160 gpurun_out/round2_fp32x3_bench.json
timeout 900 python bench.py --dp-selftest --steps 10 --warmup 3 > gpurun_out/round2_dp_selftest_bench.json 2>/dev/null; cut -c1-160 gpurun_out/round2_dp_selftest_bench.json
timeout 900 python bench.py --workload infer_rerank_beam30 --no-cpu-baseline > gpurun_out/round2_infer_bench.json 2>/dev/null; cut -c1-160 gpurun_out/round2_infer_bench.json
