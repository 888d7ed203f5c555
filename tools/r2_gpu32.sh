#!/bin/bash
# round-2 GPU pass 32: native GEMM experiment builds (scratch, selected with YTVLN_LIB) against the tree build, per shape, alternated
mkdir -p gpurun_out
{
for rep in 1 2; do
echo "== tree"; SHAPES=${SH:-all} timeout 300 python tools/gemm_shapes_bench.py 2>&1 | grep -v amdgpu.ids
for n in "$@"; do echo "== $n"; YTVLN_LIB=$PWD/scratch/epi/libepi_$n.so SHAPES=${SH:-all} timeout 300 python tools/gemm_shapes_bench.py 2>&1 | grep -v amdgpu.ids; done
done
} > gpurun_out/r2_gemm_exp.log 2>&1
python - <<'PY'
import collections, re
rows = collections.OrderedDict(); cur = None
for l in open("gpurun_out/r2_gemm_exp.log"):
    if l.startswith("=="): cur = l[3:].strip(); continue
    p = l.split()
    if len(p) >= 8 and " tA" in l:
        rows.setdefault(" ".join(p[:5]), collections.OrderedDict()).setdefault(cur, []).append(float(p[5]))
names = list(next(iter(rows.values())).keys())
print("shape".ljust(28) + "".join(n.rjust(18) for n in names))
for k, d in rows.items():
    print(k.ljust(28) + "".join(("%8.1f" % (sum(v) / len(v)) + " us").rjust(18) for v in d.values()))
PY
