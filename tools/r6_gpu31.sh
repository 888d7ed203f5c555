#!/bin/bash
# exactly what the driver runs at round end: the GPU tests with -x, smoke(), the default bench line (timed)
export TMPDIR=/tmp; mkdir -p gpurun_out
t0=$(date +%s); python -m pytest tests/ -x -q -m gpu 2>&1 | grep -a "passed\|failed" | tail -2; t1=$(date +%s); echo "pytest seconds $((t1-t0))"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1; t2=$(date +%s); echo "smoke seconds $((t2-t1))"
python bench.py > gpurun_out/r6_driver_like_bench.json 2>/dev/null; t3=$(date +%s); echo "bench seconds $((t3-t2))"
python -c "
import json; d=json.loads(open('gpurun_out/r6_driver_like_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'], len(open('gpurun_out/r6_driver_like_bench.json').read().strip().splitlines()), 'line(s)')"
