#!/bin/bash
export TMPDIR=/tmp
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --no-variants --no-cpu-baseline --no-kernel-timing --host-probe 0 --steps 20 --warmup 5 $EXTRA 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', d['value'], 'cpu', d['host_cpu_process_ms_per_step'], [ (t['cpu_ms']) for t in d['host_busiest_threads_ms_per_step'][:3]])"; }
EXTRA="--two-stream off" run one_stream X=1
EXTRA="" run default X=1
EXTRA="" run hsa_int0 HSA_ENABLE_INTERRUPT=0
EXTRA="" run maxq1 GPU_MAX_HW_QUEUES=1
EXTRA="" run dd0_maxq GPU_MAX_HW_QUEUES=4 AMD_DIRECT_DISPATCH=0
EXTRA="--graph off" run eager X=1
