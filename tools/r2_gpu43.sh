#!/bin/bash
# round-2 GPU pass 43: text-stream forward / input-gradient shapes forced onto each tile shape (planner knobs), no split
mkdir -p gpurun_out
{
echo "== planner"; SHAPES=${SH:-fwddx} timeout 300 python tools/gemm_shapes_bench.py 2>&1 | grep -v amdgpu.ids
for t in 0 1 3 4; do echo "== tile$t"; YTVLN_GEMM_TILE=$t YTVLN_GEMM_SPLITS=1 SHAPES=${SH:-fwddx} timeout 300 python tools/gemm_shapes_bench.py 2>&1 | grep -v amdgpu.ids; done
for sp in 2 3 4; do echo "== t4s$sp"; YTVLN_GEMM_TILE=4 YTVLN_GEMM_SPLITS=$sp SHAPES=${SH:-fwddx} timeout 300 python tools/gemm_shapes_bench.py 2>&1 | grep -v amdgpu.ids; done
} > gpurun_out/r2_text_tiles.log 2>&1
python - <<'PY'
import collections
rows = collections.OrderedDict(); cur = None
for l in open("gpurun_out/r2_text_tiles.log"):
    if l.startswith("=="): cur = l[3:].strip(); continue
    p = l.split()
    if len(p) >= 8 and " tA" in l:
        rows.setdefault(" ".join(p[:5]), collections.OrderedDict())[cur] = float(p[5])
names = list(next(iter(rows.values())).keys())
print("shape".ljust(26) + "".join(n.rjust(10) for n in names))
for k, d in rows.items():
    print(k.ljust(26) + "".join(("%8.1f" % v).rjust(10) for v in d.values()))
PY
