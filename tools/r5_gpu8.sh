#!/bin/bash
# A/B on one box: bf16 attention with stored keep bits (new) vs regenerated hash (old library build), micro-benchmark + cfg5 step, ABAB
mkdir -p gpurun_out; export TMPDIR=/tmp
OLD=$PWD/youtube-vln_amd/ytvln/lib/libytvln_oldattn.so
for rep in 1 2; do
  echo "== new $rep"; PRECISION=bf16 PAIRS_N=224 REGIONS=576 timeout 300 python tools/attn_bench.py 2>/dev/null
  echo "== old $rep"; YTVLN_LIB=$OLD PRECISION=bf16 PAIRS_N=224 REGIONS=576 timeout 300 python tools/attn_bench.py 2>/dev/null
done
for rep in 1 2; do
  for w in new old; do
    if [ $w = old ]; then export YTVLN_LIB=$OLD; else unset YTVLN_LIB; fi
    timeout 600 python bench.py --workload cfg5_long_traj_bs32 --precision bf16 --no-variants --no-cpu-baseline --steps 6 --warmup 2 > gpurun_out/r5g_cfg5_$w$rep.json 2>/dev/null
    python - <<PY
import json
d=json.loads(open('gpurun_out/r5g_cfg5_$w$rep.json').read().strip().splitlines()[-1]); print('CFG5 $w $rep', d['value'], d['ms_per_step'], {k:v.get('ms_per_step') for k,v in d['roofline'].get('families',{}).items()})
PY
  done
done
