#!/bin/bash
mkdir -p gpurun_out
# call 15 of the process = a steady-state launch of the first shape (3 warm-up + 20 timed per shape)
YTVLN_GEMM_DBG=15 YTVLN_GEMM_SW=1 SHAPES=img python tools/gemm_shapes_bench.py > gpurun_out/r3_dbg13.out 2> gpurun_out/r3_dbg13.err
grep -c gemmdbg gpurun_out/r3_dbg13.err; head -3 gpurun_out/r3_dbg13.out
