#!/bin/bash
# round-2 GPU pass 14: interleaved fragments for M/N-contiguous GEMM operands: parity, per-shape rates, 256x256 tiles for weight gradients
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" > gpurun_out/r2_gemm_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r2_gemm_tests.log
tail -4 gpurun_out/r2_gemm_tests.log
run() { echo "== $1"; env $1 timeout 300 python tools/gemm_shapes_bench.py 2>&1 | grep -v amdgpu.ids; }
{
run "X=default"
run "YTVLN_GEMM_BIG_TA=1"
run "YTVLN_GEMM_BIG_TA=1 YTVLN_GEMM_TILE=4 YTVLN_GEMM_SPLITS=8 SHAPES=wgrad"
run "YTVLN_GEMM_BIG_TA=1 YTVLN_GEMM_TILE=4 YTVLN_GEMM_SPLITS=16 SHAPES=wgrad"
run "YTVLN_GEMM_BIG_TA=1 YTVLN_GEMM_TILE=4 YTVLN_GEMM_SPLITS=4 SHAPES=wgrad"
run "YTVLN_GEMM_BIG_TA=1 YTVLN_GEMM_TILE=4 YTVLN_GEMM_SPLITS=2 SHAPES=wgrad"
run "YTVLN_GEMM_BIG_TA=1 YTVLN_GEMM_TILE=4 YTVLN_GEMM_SPLITS=1 SHAPES=wgrad"
} > gpurun_out/r2_gemm_shapes.log 2>&1
cat gpurun_out/r2_gemm_shapes.log
timeout 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-variants > gpurun_out/r2_bench_il.json 2>/dev/null; cut -c1-200 gpurun_out/r2_bench_il.json
YTVLN_GEMM_BIG_TA=1 timeout 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-variants > gpurun_out/r2_bench_il_bigta.json 2>/dev/null; cut -c1-200 gpurun_out/r2_bench_il_bigta.json
