#!/bin/bash
# round-2 GPU pass 2: new attention kernels -- parity tests, then the micro-benchmark over the launch-shape variants
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "attention" > gpurun_out/r2_attn_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r2_attn_tests.log
tail -15 gpurun_out/r2_attn_tests.log
timeout 600 python -m pytest tests/test_model_gpu.py tests/test_dp_gpu.py -x -q -m gpu -k "g0 or load_state_dict or two_ranks" > gpurun_out/r2_model_quick.log 2>&1; echo "rc=$?" >> gpurun_out/r2_model_quick.log
tail -6 gpurun_out/r2_model_quick.log
run() { echo "== $1" ; env $1 timeout 300 python tools/attn_bench.py 2>&1 | grep -v "^$" ; }
{
run "X=default"
run "YTVLN_ATTN_WAVES=4"
run "YTVLN_ATTN_WAVES=3"
run "YTVLN_ATTN_WAVES=1"
run "YTVLN_ATTN_PAIRS=2 YTVLN_ATTN_DKV_STAGES=2"
run "YTVLN_ATTN_PAIRS=1 YTVLN_ATTN_DKV_STAGES=2"
run "YTVLN_ATTN_PAIRS=2 YTVLN_ATTN_DKV_STAGES=1"
run "PDROP=0.0"
} > gpurun_out/r2_attn_bench.log 2>&1
cat gpurun_out/r2_attn_bench.log
