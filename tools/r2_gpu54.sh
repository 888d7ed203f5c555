#!/bin/bash
# round-2 GPU pass 54: 128x128 tiles with a 3- / 4-stage operand ring at one workgroup per CU (scratch build, YTVLN_GEMM_CFG) on the text shapes
mkdir -p gpurun_out
L=$PWD/scratch/epi/libepi_ns4.so
{
echo "== planner"; SHAPES=fwddx timeout 300 python tools/gemm_shapes_bench.py 2>&1 | grep -v amdgpu.ids
echo "== t0(2st,2wg)"; YTVLN_LIB=$L YTVLN_GEMM_TILE=0 YTVLN_GEMM_SPLITS=1 SHAPES=fwddx timeout 300 python tools/gemm_shapes_bench.py 2>&1 | grep -v amdgpu.ids
echo "== t0 4-stage"; YTVLN_LIB=$L YTVLN_GEMM_CFG=3 YTVLN_GEMM_TILE=0 YTVLN_GEMM_SPLITS=1 SHAPES=fwddx timeout 300 python tools/gemm_shapes_bench.py 2>&1 | grep -v amdgpu.ids
echo "== t0 3-stage"; YTVLN_LIB=$L YTVLN_GEMM_CFG=4 YTVLN_GEMM_TILE=0 YTVLN_GEMM_SPLITS=1 SHAPES=fwddx timeout 300 python tools/gemm_shapes_bench.py 2>&1 | grep -v amdgpu.ids
echo "== t0 64-deep"; YTVLN_LIB=$L YTVLN_GEMM_CFG=2 YTVLN_GEMM_TILE=0 YTVLN_GEMM_SPLITS=1 SHAPES=fwddx timeout 300 python tools/gemm_shapes_bench.py 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r2_ring.log 2>&1
python - <<'PY'
import collections
rows = collections.OrderedDict(); cur = None
for l in open("gpurun_out/r2_ring.log"):
    if l.startswith("=="): cur = l[3:].strip(); continue
    p = l.split()
    if len(p) >= 8 and " tA" in l:
        rows.setdefault(" ".join(p[:5]), collections.OrderedDict())[cur] = float(p[5])
names = list(next(iter(rows.values())).keys())
print("shape".ljust(26) + "".join(n[:13].rjust(14) for n in names))
for k, d in rows.items():
    print(k.ljust(26) + "".join(("%9.1f" % v).rjust(14) for v in d.values()))
PY
