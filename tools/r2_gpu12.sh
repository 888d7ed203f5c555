#!/bin/bash
# round-2 GPU pass 12: full suite on the final tree (incl. g13 branches), smoke
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/round2_gpu_tests.log 2>&1; echo "rc=$?" >> gpurun_out/round2_gpu_tests.log
tail -4 gpurun_out/round2_gpu_tests.log
python __graft_entry__.py smoke 2>&1 | tail -1
