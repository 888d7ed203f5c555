#!/bin/bash
# round-2 GPU pass 10: d-split single-tile workgroups in the forward kernel: parity, micro-benchmark A/B, counters
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "attention or g0 or g2_full_model_all or graph_replay" > gpurun_out/r2_ds_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r2_ds_tests.log
tail -4 gpurun_out/r2_ds_tests.log
{ echo "== dsplit on"; timeout 300 python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids;
  echo "== dsplit off"; YTVLN_ATTN_DSPLIT=0 timeout 300 python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids;
  echo "== delta kernel"; YTVLN_ATTN_DELTA_KERNEL=1 CASES=img timeout 300 python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids; } > gpurun_out/r2_ds_bench.log 2>&1
cat gpurun_out/r2_ds_bench.log
bash tools/attn_pmc.sh r2ds img > gpurun_out/r2ds_attn_pmc.txt 2>&1; tail -4 gpurun_out/r2ds_attn_pmc.txt | cut -c1-220
timeout 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-variants > gpurun_out/r2_bench_ds.json 2>/dev/null; cut -c1-200 gpurun_out/r2_bench_ds.json
python __graft_entry__.py smoke 2>&1 | tail -2
