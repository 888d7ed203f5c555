#!/bin/bash
mkdir -p gpurun_out
{
  YTVLN_LIB=scratch/lib_w1t.so YTVLN_ATTN_W1=0 YTVLN_ATTN_W1_DQ=1 CASES=img timeout 300 python tools/attn_bench.py 2>&1 | grep "w1 dq timing\|img self" | tail -5
} > gpurun_out/attn_w1t.log 2>&1
cat gpurun_out/attn_w1t.log
