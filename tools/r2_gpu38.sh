#!/bin/bash
# round-2 GPU pass 38: planner with native 256x256 split-K weight-gradient tiles (tree build) against YTVLN_GEMM_BIG_TA=0, alternated
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" 2>&1 | tail -2
{
for rep in 1 2; do
echo "== big_ta=0"; YTVLN_GEMM_BIG_TA=0 SHAPES=wgrad timeout 300 python tools/gemm_shapes_bench.py 2>&1 | grep -v amdgpu.ids
echo "== big_ta=1"; SHAPES=wgrad timeout 300 python tools/gemm_shapes_bench.py 2>&1 | grep -v amdgpu.ids
done
} > gpurun_out/r2_bigta2.log 2>&1
python - <<'PY'
import collections
rows = collections.OrderedDict(); cur = None
for l in open("gpurun_out/r2_bigta2.log"):
    if l.startswith("=="): cur = l[3:].strip(); continue
    p = l.split()
    if len(p) >= 8 and " tA" in l:
        rows.setdefault(" ".join(p[:3]), collections.OrderedDict()).setdefault(cur, []).append(float(p[5]))
names = list(next(iter(rows.values())).keys())
print("shape".ljust(18) + "".join(n.rjust(14) for n in names))
for k, d in rows.items():
    print(k.ljust(18) + "".join(("%9.1f" % (sum(v) / len(v))).rjust(14) for v in d.values()))
PY
for v in 0 1; do YTVLN_GEMM_BIG_TA=$v timeout 900 python bench.py --no-cpu-baseline --no-variants --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][0]); print('big_ta=$v', d['value'], d['ms_per_step'], d['final_loss'])"; done
for v in 0 1; do YTVLN_GEMM_BIG_TA=$v timeout 900 python bench.py --no-cpu-baseline --no-variants --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][0]); print('big_ta=$v', d['value'], d['ms_per_step'], d['final_loss'])"; done
