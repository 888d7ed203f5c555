#!/bin/bash
# round-2 GPU pass 18: fp32x3 GEMM experiments (scratch builds selected with YTVLN_LIB), per-shape rates
mkdir -p gpurun_out
{
echo "== x3 current"; PRECISION=fp32x3 timeout 300 python tools/gemm_shapes_bench.py
for n in 1 2 3; do echo "== x3 exp $n"; YTVLN_LIB=$PWD/scratch/x3/libx3_$n.so PRECISION=fp32x3 timeout 300 python tools/gemm_shapes_bench.py; done
} > gpurun_out/r2_x3_exp.log 2>&1
cat gpurun_out/r2_x3_exp.log
