#!/bin/bash
# final-tree robustness records: full GPU suite, soak of the captured step (fp32 cfg 2, bf16 cfg 5), 300-step loss trajectory in three arithmetics
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 3000 python -m pytest tests -q -m gpu > gpurun_out/r5_gpu_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r5_gpu_tests.log; tail -4 gpurun_out/r5_gpu_tests.log
timeout 900 python tools/soak.py 800 round5 2>&1 | tail -3
timeout 900 python tools/soak.py 300 round5_cfg5_bf16 bf16 cfg5_long_traj_bs32 2>&1 | tail -3
timeout 900 python tools/loss_trajectory.py 300 2>&1 | tail -5
ls gpurun_out | tail -20
