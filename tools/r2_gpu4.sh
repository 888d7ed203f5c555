#!/bin/bash
# round-2 GPU pass 4: attention after read batching / cheaper DMA addressing; whole-step kernel trace; checkpoint interop
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "attention" > gpurun_out/r2_attn_tests2.log 2>&1; echo "rc=$?" >> gpurun_out/r2_attn_tests2.log
tail -3 gpurun_out/r2_attn_tests2.log
timeout 600 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "reference_written or pretrained_model or load_state_dict or dropout_stream or save_resume" > gpurun_out/r2_ckpt_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r2_ckpt_tests.log
tail -12 gpurun_out/r2_ckpt_tests.log
timeout 300 python tools/write_repo_ckpt.py 2>&1 | tail -2
{ echo "== default"; timeout 300 python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids; } > gpurun_out/r2_attn_bench2.log 2>&1
cat gpurun_out/r2_attn_bench2.log
TOPN=40 bash tools/kernel_stats.sh r2a > gpurun_out/r2a_kernel_stats.txt 2>&1
cat gpurun_out/r2a_kernel_stats.txt
