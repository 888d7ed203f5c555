#!/bin/bash
# round-2 GPU pass 16: 300-step loss trajectories (fp32 / fp32x3 / bf16); the data-parallel path at full size with 2 ranks sharing the GPU (gloo)
mkdir -p gpurun_out
timeout 1500 python tools/loss_trajectory.py 300 2>&1 | grep -v amdgpu.ids | tail -5
YTVLN_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 6 --warmup 2 > gpurun_out/round2_dp2_gloo_selftest_bench.json 2> gpurun_out/round2_dp2.err; cut -c1-700 gpurun_out/round2_dp2_gloo_selftest_bench.json; tail -3 gpurun_out/round2_dp2.err
