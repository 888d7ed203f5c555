#!/bin/bash
# round-6 soak on the final library: 1500 replays of the fp32 headline step, 400 of cfg 5 bf16
export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1200 python tools/soak.py 1500 round6 fp32 > gpurun_out/round6_soak.out 2>&1; tail -3 gpurun_out/round6_soak.out
timeout 1200 python tools/soak.py 400 round6_cfg5_bf16 bf16 cfg5_long_traj_bs32 > gpurun_out/round6_cfg5_soak.out 2>&1; tail -3 gpurun_out/round6_cfg5_soak.out
