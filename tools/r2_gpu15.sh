#!/bin/bash
# round-2 GPU pass 15: A/B of the interleaved-fragment GEMM against the previous build (scratch/lib_preil.so), same box, alternating
mkdir -p gpurun_out
b() { env $1 timeout 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-variants 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['roofline']['frac'])"; }
{
for i in 1 2 3; do
b "YTVLN_LIB=scratch/lib_preil.so"
b "X=il_default"
b "YTVLN_GEMM_BIG_TA=1"
done
echo "== preil shapes"; YTVLN_LIB=scratch/lib_preil.so timeout 300 python tools/gemm_shapes_bench.py 2>&1 | grep -v amdgpu.ids
echo "== il shapes"; timeout 300 python tools/gemm_shapes_bench.py 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r2_il_ab.log 2>&1
cat gpurun_out/r2_il_ab.log
