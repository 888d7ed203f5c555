#!/bin/bash
# round 6: full GPU test suite, then the evidence set on the same box: default bench line, one-stream kernel traces of the headline and of
# cfg5 bf16, PMC passes for the GEMM family, the K = 1 reading of "bs 8", cfg 1 / cfg 4 / inference records
set -u
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -q -m gpu > gpurun_out/r6_gpu_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r6_gpu_tests.log
tail -6 gpurun_out/r6_gpu_tests.log
timeout 1200 python bench.py > gpurun_out/round6_bench_default.json 2> gpurun_out/round6_bench_default.err
tail -c 400 gpurun_out/round6_bench_default.json; echo
cd $GRAFT_REPO_ROOT; bash tools/kernel_stats.sh round6 --graph off --two-stream off --steps 12 --warmup 3 --no-cpu-baseline --no-variants --no-kernel-timing --host-probe 0
cd $GRAFT_REPO_ROOT; bash tools/kernel_stats.sh round6_cfg5_bf16 --workload cfg5_long_traj_bs32 --precision bf16 --graph off --two-stream off --steps 6 --warmup 2 --no-cpu-baseline --no-variants --no-kernel-timing --host-probe 0
cd $GRAFT_REPO_ROOT; PMC_TAG=round6_ PMC_BENCH_ARGS="" bash tools/pmc_bench.sh > gpurun_out/round6_pmc.log 2>&1; tail -c 600 gpurun_out/round6_pmc.log; echo
cd $GRAFT_REPO_ROOT; timeout 600 python bench.py --workload cfg2_k1_rows8 --no-variants --no-cpu-baseline > gpurun_out/round6_cfg2_k1_rows8_bench.json 2> gpurun_out/round6_cfg2_k1_rows8_bench.err; tail -c 300 gpurun_out/round6_cfg2_k1_rows8_bench.json; echo
cd $GRAFT_REPO_ROOT; timeout 600 python bench.py --workload cfg1_tiny_mlm_bs2 --no-variants --no-cpu-baseline > gpurun_out/round6_cfg1_bench.json 2>/dev/null; tail -c 200 gpurun_out/round6_cfg1_bench.json; echo
cd $GRAFT_REPO_ROOT; timeout 600 python bench.py --workload cfg4_finetune_rank_bs16 --no-variants --no-cpu-baseline > gpurun_out/round6_cfg4_bench.json 2>/dev/null; tail -c 200 gpurun_out/round6_cfg4_bench.json; echo
cd $GRAFT_REPO_ROOT; timeout 900 python tools/gemm_vs_torch.py > gpurun_out/round6_gemm_vs_library_unfair_order.log 2>&1
cd $GRAFT_REPO_ROOT; SHAPES=all3 CONFIGS=old timeout 900 python tools/gemm_sk_bench.py > gpurun_out/round6_gemm_vs_library.log 2>&1; cat gpurun_out/round6_gemm_vs_library.log
