#!/bin/bash
mkdir -p gpurun_out
L=gpurun_out/r3_sw12.log
: > $L
for M in plain bias gelu beta dgelu; do for C in 0 6; do for V in "X=default" "YTVLN_GEMM_SW=1" "YTVLN_GEMM_SW=1" "X=default"; do
  echo "== MODE=$M COLD=$C $V" >> $L
  env $V MODE=$M COLD=$C SHAPES=img timeout 600 python tools/gemm_shapes_bench.py 2>/dev/null >> $L
done; done; done
cat $L
