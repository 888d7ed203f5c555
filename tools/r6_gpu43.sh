#!/bin/bash
# bf16 attention backward with the exponent built in base 2 and the dropout rescale outside the loop, against the build before (libytvln_prevattn.so):
# bf16 tests, the attention sites of cfg 5 alone, cfg 5 ABAB
export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_bf16_gpu.py tests/test_model_gpu.py -m gpu -x -q -k "bf16" 2>&1 | grep -a "passed\|failed" | tail -2
{
for rep in 1 2; do for v in new prev; do
if [ $v = new ]; then unset YTVLN_LIB; else export YTVLN_LIB=$PWD/youtube-vln_amd/ytvln/lib/libytvln_prevattn.so; fi
echo "== $v (pass $rep)"; PRECISION=bf16 PAIRS_N=224 REGIONS=576 timeout 600 python tools/attn_bench.py 2>/dev/null | cut -c1-110
done; done
for rep in 1 2; do for v in new prev; do
if [ $v = new ]; then unset YTVLN_LIB; else export YTVLN_LIB=$PWD/youtube-vln_amd/ytvln/lib/libytvln_prevattn.so; fi
timeout 900 python bench.py --workload cfg5_long_traj_bs32 --precision bf16 --no-variants --no-cpu-baseline --steps 6 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=d['roofline']['families']; print('CFG5 $v', d['value'], d['ms_per_step'], 'gemm', f['gemm']['ms_per_step'], 'attn', f['attention']['ms_per_step'], f['attention']['frac'], 'loss', d['final_loss'])"
done; done
} > gpurun_out/r6_battn_exp2_ab.log 2>&1
cat gpurun_out/r6_battn_exp2_ab.log
