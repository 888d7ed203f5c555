#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
for cfg in "4 0" "4 1" "0 0" "0 1"; do set -- $cfg
echo "== GEMM_TILE=$1 GEMM_BF16_PERSIST=$2"; YTVLN_GEMM_TILE=$1 YTVLN_GEMM_BF16_PERSIST=$2 timeout 600 python tools/gemm_bf16_ksweep.py 2>&1 | grep -v amdgpu.ids
done
