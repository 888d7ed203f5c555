#!/bin/bash
# round-2 GPU pass 8: image layers beside text layers on a second HIP stream (YTVLN_DUAL_STREAM=1): parity + bench A/B
mkdir -p gpurun_out
YTVLN_DUAL_STREAM=1 timeout 1200 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "g0 or g2_full_model_all or graph_replay or g11 or dropout" > gpurun_out/r2_dual_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r2_dual_tests.log
tail -5 gpurun_out/r2_dual_tests.log
for i in 1 2; do
timeout 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-variants --no-kernel-timing > gpurun_out/r2_bench_single_$i.json 2> gpurun_out/r2_bench_single_$i.err; cut -c1-200 gpurun_out/r2_bench_single_$i.json
YTVLN_DUAL_STREAM=1 timeout 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-variants --no-kernel-timing > gpurun_out/r2_bench_dual_$i.json 2> gpurun_out/r2_bench_dual_$i.err; cut -c1-200 gpurun_out/r2_bench_dual_$i.json; tail -2 gpurun_out/r2_bench_dual_$i.err
done
