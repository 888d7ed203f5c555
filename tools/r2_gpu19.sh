#!/bin/bash
# round-2 GPU pass 19: fp32x3 GEMM timing probes (scratch builds), big shapes only
mkdir -p gpurun_out
{
for n in "$@"; do echo "== x3 $n"; YTVLN_LIB=$PWD/scratch/x3/libx3_$n.so PRECISION=fp32x3 SHAPES=${SH:-fwd} timeout 300 python tools/gemm_shapes_bench.py 2>&1 | grep -v amdgpu.ids; done
} > gpurun_out/r2_x3_probe.log 2>&1
cat gpurun_out/r2_x3_probe.log
