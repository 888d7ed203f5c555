#!/bin/bash
# branch-free Phi in the bf16 GEMM epilogues against the build with the library's erff (libytvln_erff.so): bf16 tests, the epilogue costs, cfg 5 ABAB
export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_bf16_gpu.py tests/test_model_gpu.py -m gpu -x -q -k "bf16" 2>&1 | grep -a "passed\|failed" | tail -2
{
for v in new erff; do
if [ $v = new ]; then unset YTVLN_LIB; else export YTVLN_LIB=$PWD/youtube-vln_amd/ytvln/lib/libytvln_erff.so; fi
echo "== $v"; timeout 600 python tools/gemm_bf16_epilogues.py 2>/dev/null
done
for rep in 1 2; do for v in new erff; do
if [ $v = new ]; then unset YTVLN_LIB; else export YTVLN_LIB=$PWD/youtube-vln_amd/ytvln/lib/libytvln_erff.so; fi
timeout 900 python bench.py --workload cfg5_long_traj_bs32 --precision bf16 --no-variants --no-cpu-baseline --steps 6 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=d['roofline']['families']; print('CFG5 $v', d['value'], d['ms_per_step'], 'gemm', f['gemm']['ms_per_step'], f['gemm']['frac'], 'attn', f['attention']['ms_per_step'], 'loss', d['final_loss'])"
done; done
} > gpurun_out/r6_bf16_phi_ab.log 2>&1
cat gpurun_out/r6_bf16_phi_ab.log
