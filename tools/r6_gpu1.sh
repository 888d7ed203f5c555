#!/bin/bash
# round 6, call 1: ABI v2 loads; the new bf16 GEMM main-loop forms (DMA issue inside the matrix phases) and the first-round stagger -- correctness
# on the full grids, then the per-shape / K-sweep A/B against the shipped form and the library; host footprint of the headline, blocking vs spin
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_abi.py -x -q 2>&1 | tail -2
for f in 1 2; do
  YTVLN_GEMM_BF16_FORM=$f timeout 900 python -m pytest tests/test_bf16_gpu.py -m gpu -x -q -k "gemm" 2>&1 | tail -2
done
YTVLN_GEMM_BF16_FORM=2 YTVLN_GEMM_STAGGER=50 timeout 900 python -m pytest tests/test_bf16_gpu.py -m gpu -x -q -k "gemm_bf16_full_grids" 2>&1 | tail -2
KSWEEP=1 timeout 1200 python tools/gemm_bf16_forms.py > gpurun_out/r6_gemm_bf16_forms.log 2>&1; cat gpurun_out/r6_gemm_bf16_forms.log
for hw in blocking spin; do
  timeout 600 python bench.py --no-variants --no-cpu-baseline --steps 20 --warmup 5 --host-wait $hw > gpurun_out/r6_bench_$hw.json 2> gpurun_out/r6_bench_$hw.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r6_bench_$hw.json').read().strip().splitlines()[-1])
print('HEADLINE $hw', d['value'], d['ms_per_step'], 'host_cpu_process', d['host_cpu_process_ms_per_step'], 'enqueue', d['host_enqueue_ms_per_step'], d.get('host_busiest_threads_ms_per_step'), d['roofline']['frac'])
PY
done
timeout 300 python tools/find_stray_grads.py > gpurun_out/r6_stray.log 2>&1; tail -45 gpurun_out/r6_stray.log
timeout 300 python tools/find_copies2.py > gpurun_out/r6_copies.log 2>&1; head -70 gpurun_out/r6_copies.log
