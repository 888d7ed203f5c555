#!/bin/bash
# round-2 GPU pass 21: split-major XCD layout of split-K workgroups -- parity, per-shape rate A/B, traffic counters
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" 2>&1 | tail -2
{
echo "== split_map=0"; YTVLN_GEMM_SPLIT_MAP=0 SHAPES=wgrad timeout 300 python tools/gemm_shapes_bench.py 2>&1 | grep -v amdgpu.ids
echo "== split_map=1"; SHAPES=wgrad timeout 300 python tools/gemm_shapes_bench.py 2>&1 | grep -v amdgpu.ids
echo "== split_map=0 x3"; PRECISION=fp32x3 YTVLN_GEMM_SPLIT_MAP=0 SHAPES=wgrad timeout 300 python tools/gemm_shapes_bench.py 2>&1 | grep -v amdgpu.ids
echo "== split_map=1 x3"; PRECISION=fp32x3 SHAPES=wgrad timeout 300 python tools/gemm_shapes_bench.py 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r2_splitmap.log 2>&1
cat gpurun_out/r2_splitmap.log
bash tools/gemm_pmc.sh r2b | cut -c1-330
