#!/bin/bash
mkdir -p gpurun_out
L=gpurun_out/r3_clk14.log
: > $L
for V in "X=default" "YTVLN_GEMM_SW=1" "YTVLN_GEMM_SW=1 YTVLN_GEMM_PROBE=1" "YTVLN_GEMM_SW=1 YTVLN_GEMM_PROBE=2" "YTVLN_GEMM_SW=1 YTVLN_GEMM_PROBE=3" "YTVLN_GEMM_TILE=0" "YTVLN_GEMM_TILE=0 YTVLN_GEMM_SW=1" "YTVLN_GEMM_TILE=3 YTVLN_GEMM_SW=1" "X=default" "YTVLN_GEMM_SW=1"; do
  for S in "0 1" ; do
  echo "== $V" >> $L
  env $V YTVLN_GEMM_DBG=15 SHAPES=${SH:-img1} timeout 600 python tools/gemm_shapes_bench.py 2>&1 | grep -v amdgpu.ids >> $L
  done
done
cat $L
