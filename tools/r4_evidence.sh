#!/bin/bash
# end-of-round evidence on one box: default bench line, kernel traces of the headline and of cfg5 bf16 (one stream: kernels alone on the chip),
# reproducibility of the bf16 kernels under concurrency, loss trajectories
set -u
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p gpurun_out
timeout 1200 python bench.py > gpurun_out/round4b_bench_default.json 2> gpurun_out/round4b_bench_default.err
tail -c 300 gpurun_out/round4b_bench_default.json; echo
cd $GRAFT_REPO_ROOT; bash tools/kernel_stats.sh round4b --graph off --two-stream off --steps 12 --warmup 3 --no-cpu-baseline --no-variants --no-kernel-timing --host-probe 0
cd $GRAFT_REPO_ROOT; bash tools/kernel_stats.sh round4b_cfg5_bf16 --workload cfg5_long_traj_bs32 --precision bf16 --graph off --two-stream off --steps 6 --warmup 2 --no-cpu-baseline --no-variants --no-kernel-timing --host-probe 0
cd $GRAFT_REPO_ROOT; timeout 900 python tools/bf16_repro.py 30 > gpurun_out/round4_bf16_repro.log 2>&1; grep -c "alone 0/30  concurrent 0/30" gpurun_out/round4_bf16_repro.log; grep -v "alone 0/30  concurrent 0/30" gpurun_out/round4_bf16_repro.log | tail -5
timeout 1500 python tools/loss_trajectory.py 300 round4b > gpurun_out/round4b_traj.log 2>&1; tail -3 gpurun_out/round4b_traj.log
