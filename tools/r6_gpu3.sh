#!/bin/bash
# round 6, call 3: phase timeline of the bf16 GEMM main loop; host footprint with the bench's own sampler thread accounted; the full GPU suite
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python tools/gemm_bf16_probe.py > gpurun_out/r6_gemm_bf16_probe.log 2>&1; cat gpurun_out/r6_gemm_bf16_probe.log
for hw in blocking spin; do
  timeout 600 python bench.py --no-variants --no-cpu-baseline --no-kernel-timing --host-probe 0 --steps 20 --warmup 5 --host-wait $hw > gpurun_out/r6_host_$hw.json 2> gpurun_out/r6_host_$hw.err || tail -5 gpurun_out/r6_host_$hw.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r6_host_$hw.json').read().strip().splitlines()[-1])
print('HOST $hw:', d['value'], d['ms_per_step'], 'process cpu', d['host_cpu_process_ms_per_step'], 'sampler', d['bench_power_sampler_cpu_ms_per_step'], d['host_thread_cpu_ms_per_step'], d.get('host_busiest_threads_ms_per_step'))
PY
done
timeout 600 python bench.py --no-variants --no-cpu-baseline --no-kernel-timing --host-probe 0 --steps 10 --warmup 3 --loop reference > gpurun_out/r6_refloop.json 2> gpurun_out/r6_refloop.err || tail -5 gpurun_out/r6_refloop.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r6_refloop.json').read().strip().splitlines()[-1])
print('REFERENCE LOOP:', d['value'], d['ms_per_step'], d['config']['execution'])
PY
timeout 2400 python -m pytest tests -q -m gpu -x > gpurun_out/r6_gpu_tests.log 2>&1; tail -8 gpurun_out/r6_gpu_tests.log
