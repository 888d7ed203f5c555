#!/bin/bash
# round 6, call 2: bf16 GEMM form 3 (32-deep tiles, five-slot rings, three tiles in flight): correctness on the full grids, A/B + K sweep against
# the shipped form and the library; host pacing matrix of the headline (event / poll, depth, one or two executable graphs)
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
YTVLN_GEMM_BF16_FORM=3 timeout 900 python -m pytest tests/test_bf16_gpu.py -m gpu -x -q -k "gemm" 2>&1 | tail -3
CONFIGS=0:0,3:0 KSWEEP=1 timeout 1200 python tools/gemm_bf16_forms.py > gpurun_out/r6_gemm_bf16_form3.log 2>&1; cat gpurun_out/r6_gemm_bf16_form3.log
for cfg in "event:2 1" "event:1 2" "poll:1 2" "poll:2 1" "poll:1 1"; do
  set -- $cfg
  timeout 600 python bench.py --no-variants --no-cpu-baseline --no-kernel-timing --host-probe 0 --steps 20 --warmup 5 --pace $1 --graphs $2 > gpurun_out/r6_pace.json 2> gpurun_out/r6_pace.err || tail -5 gpurun_out/r6_pace.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r6_pace.json').read().strip().splitlines()[-1])
print('PACE $1 graphs $2:', d['value'], d['ms_per_step'], 'process cpu', d['host_cpu_process_ms_per_step'], d['host_thread_cpu_ms_per_step'], d.get('host_busiest_threads_ms_per_step')[:2])
PY
done
