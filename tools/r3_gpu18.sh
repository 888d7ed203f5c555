#!/bin/bash
mkdir -p gpurun_out
L=gpurun_out/r3_clk18.log
: > $L
for SH in img1 big1; do for D in randn zeros ones randn; do for V in "X=default" "YTVLN_GEMM_SW=1"; do
  echo "== $SH DATA=$D $V" >> $L
  env $V DATA=$D YTVLN_GEMM_DBG=15 SHAPES=$SH timeout 600 python tools/gemm_shapes_bench.py 2>&1 | grep -v amdgpu.ids >> $L
done; done; done
cat $L
