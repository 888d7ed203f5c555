#!/bin/bash
# ABBA end-to-end comparison of two environment settings: tools/r3_abba.sh "<env A>" "<env B>" [extra bench args]
mkdir -p gpurun_out
A="$1"; B="$2"; shift 2
for V in "$A" "$B" "$B" "$A"; do
  env $V timeout 900 python bench.py --no-variants --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$V', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'])"
done
