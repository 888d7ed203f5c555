#!/bin/bash
# planner: split-K of 128x128 tiles priced up for K-contiguous A x k-major B (input gradients).  The cfg-4 shapes in isolation, GEMM tests,
# cfg 4 and the headline against the library before (libytvln_prevbk.so), ABAB
export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_gemm_sk_gpu.py tests/test_abi.py -m gpu -x -q 2>&1 | grep -a "passed\|failed" | tail -1
{
SHAPE="7680,768,3072,0;7680,768,2304,0;7680,768,2048,0;7680,768,1024,0;4480,768,3072,0;4480,768,2048,0" CONFIGS=old,TS0_1,TS0_2,TS5_2,TS6_2 timeout 900 python tools/gemm_sk_bench.py 2>/dev/null | cut -c1-200
for rep in 1 2; do for v in new prev; do
if [ $v = new ]; then unset YTVLN_LIB; else export YTVLN_LIB=$PWD/youtube-vln_amd/ytvln/lib/libytvln_prevbk.so; fi
timeout 600 python bench.py --workload cfg4_finetune_rank_bs16 --no-variants --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=d['roofline']['families']; print('CFG4 $v', d['value'], d['ms_per_step'], 'gemm', f['gemm']['ms_per_step'], f['gemm']['frac'])"
done; done
unset YTVLN_LIB
bash tools/r6_gpu11.sh base prevbk
} > gpurun_out/r6_planner_bk_ab.log 2>&1
cat gpurun_out/r6_planner_bk_ab.log
