#!/bin/bash
# round-2 GPU pass 7: remaining config tests, dK/dV kernel at 36 KB of LDS, full suite
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/r2_full_tests2.log 2>&1; echo "rc=$?" >> gpurun_out/r2_full_tests2.log
tail -8 gpurun_out/r2_full_tests2.log
{ echo "== default"; timeout 300 python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids; 
  echo "== PAIRS=2 STAGES=2"; YTVLN_ATTN_PAIRS=2 YTVLN_ATTN_DKV_STAGES=2 CASES=img timeout 300 python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids; } > gpurun_out/r2_attn_bench3.log 2>&1
cat gpurun_out/r2_attn_bench3.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2_bench2.json 2> gpurun_out/r2_bench2.err; cut -c1-260 gpurun_out/r2_bench2.json
