#!/bin/bash
mkdir -p gpurun_out
L=gpurun_out/r3_sk19.log
: > $L
echo "== tests STREAMK=3" >> $L
YTVLN_GEMM_STREAMK=3 timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm and not x3 and not bf16 and not streamk_opt" 2>&1 | tail -3 >> $L
for V in "X=default" "YTVLN_GEMM_STREAMK=3" "YTVLN_GEMM_STREAMK=2" "YTVLN_GEMM_STREAMK=3" "X=default"; do
  echo "== $V" >> $L
  env $V SHAPES=fwddx timeout 600 python tools/gemm_shapes_bench.py 2>/dev/null >> $L
done
cat $L
