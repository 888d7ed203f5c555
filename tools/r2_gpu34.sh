#!/bin/bash
# round-2 GPU pass 34: fp32x3 256x256 kernel with the one-tile lookahead (scratch build) -- parity tests through YTVLN_LIB, per-shape A/B
mkdir -p gpurun_out
L=$PWD/scratch/x3/libx3_${1:-la}.so
YTVLN_LIB=$L timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "x3 or gemm" 2>&1 | tail -3
{
for rep in 1 2; do
echo "== tree"; PRECISION=fp32x3 SHAPES=${SH:-all} timeout 300 python tools/gemm_shapes_bench.py 2>&1 | grep -v amdgpu.ids
echo "== lookahead"; YTVLN_LIB=$L PRECISION=fp32x3 SHAPES=${SH:-all} timeout 300 python tools/gemm_shapes_bench.py 2>&1 | grep -v amdgpu.ids
done
} > gpurun_out/r2_x3_la.log 2>&1
python - <<'PY'
import collections
rows = collections.OrderedDict(); cur = None
for l in open("gpurun_out/r2_x3_la.log"):
    if l.startswith("=="): cur = l[3:].strip(); continue
    p = l.split()
    if len(p) >= 8 and " tA" in l:
        rows.setdefault(" ".join(p[:5]), collections.OrderedDict()).setdefault(cur, []).append(float(p[5]))
names = list(next(iter(rows.values())).keys())
print("shape".ljust(28) + "".join(n.rjust(18) for n in names))
for k, d in rows.items():
    print(k.ljust(28) + "".join(("%8.1f" % (sum(v) / len(v)) + " us").rjust(18) for v in d.values()))
PY
