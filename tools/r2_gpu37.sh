#!/bin/bash
# round-2 GPU pass 37: 256x256 split-K tiles for the native weight-gradient GEMMs (scratch build), forced tile / split combinations
mkdir -p gpurun_out
L=$PWD/scratch/epi/libepi_bt.so
{
echo "== tree"; SHAPES=wgrad timeout 300 python tools/gemm_shapes_bench.py 2>&1 | grep -v amdgpu.ids
echo "== big_ta(planner)"; YTVLN_LIB=$L YTVLN_GEMM_BIG_TA=1 SHAPES=wgrad timeout 300 python tools/gemm_shapes_bench.py 2>&1 | grep -v amdgpu.ids
for sp in 16 8 6 5 4 3; do echo "== t4s$sp"; YTVLN_LIB=$L YTVLN_GEMM_BIG_TA=1 YTVLN_GEMM_TILE=4 YTVLN_GEMM_SPLITS=$sp SHAPES=wgrad timeout 300 python tools/gemm_shapes_bench.py 2>&1 | grep -v amdgpu.ids; done
echo "== tree2"; SHAPES=wgrad timeout 300 python tools/gemm_shapes_bench.py 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r2_bigta.log 2>&1
python - <<'PY'
import collections
rows = collections.OrderedDict(); cur = None
for l in open("gpurun_out/r2_bigta.log"):
    if l.startswith("=="): cur = l[3:].strip(); continue
    p = l.split()
    if len(p) >= 8 and " tA" in l:
        rows.setdefault(" ".join(p[:3]), collections.OrderedDict())[cur] = float(p[5])
names = list(next(iter(rows.values())).keys())
print("shape".ljust(18) + "".join(n[:10].rjust(11) for n in names))
for k, d in rows.items():
    print(k.ljust(18) + "".join(("%9.1f" % v).rjust(11) for v in d.values()))
PY
