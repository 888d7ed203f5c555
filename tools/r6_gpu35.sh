#!/bin/bash
# split-K reduce with 4 / 8 partial tiles loaded per batch (libytvln_u4.so / _u8.so) against the shipped loop: weight-gradient shapes, then the headline
export TMPDIR=/tmp; mkdir -p gpurun_out
{
for rep in 1 2; do for v in base u4 u8; do
if [ $v = base ]; then unset YTVLN_LIB; else export YTVLN_LIB=$PWD/youtube-vln_amd/ytvln/lib/libytvln_$v.so; fi
echo "== $v (pass $rep)"; SHAPES=dw CONFIGS=old timeout 600 python tools/gemm_sk_bench.py 2>/dev/null | cut -c1-60
done; done
bash tools/r6_gpu11.sh base u4 u8
} > gpurun_out/r6_splitk_reduce_unroll_ab.log 2>&1
cat gpurun_out/r6_splitk_reduce_unroll_ab.log
