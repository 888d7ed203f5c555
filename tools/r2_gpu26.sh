#!/bin/bash
# round-2 GPU pass 26: sorted scatter-add spread over column chunks -- its tests, the model goldens that use it, headline bench + trace
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "scatter or embed or g0 or g2_full_model_all or replay" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/round2_bench.json 2> gpurun_out/round2_bench.err; cut -c1-240 gpurun_out/round2_bench.json
TOPN=45 bash tools/kernel_stats.sh round2 > gpurun_out/round2_kernel_stats.txt 2>&1; grep -n "scatter\|total kernel" gpurun_out/round2_kernel_stats.txt | cut -c1-160
