#!/bin/bash
# headline A/B of alternate library builds: tools/r6_gpu11.sh <name> ... (libytvln_<name>.so; "base" = the shipped library), interleaved, 2 passes
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2; do for v in "$@"; do
if [ $v = base ]; then unset YTVLN_LIB; else export YTVLN_LIB=$PWD/youtube-vln_amd/ytvln/lib/libytvln_$v.so; fi
timeout 600 python bench.py --no-variants --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r6_ab_$v.json 2>/dev/null
python - <<PY
import json
d=json.loads(open('gpurun_out/r6_ab_$v.json').read().strip().splitlines()[-1]); f=d['roofline']['families']['gemm']; fa=d['roofline']['families']['attention']; print('HEADLINE $v', d['value'], d['ms_per_step'], 'gemm ms', f['ms_per_step'], f['frac'], 'attention ms', fa['ms_per_step'], fa['frac'])
PY
done; done
