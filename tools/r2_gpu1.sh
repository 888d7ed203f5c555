#!/bin/bash
# round-2 GPU pass 1: RCCL / DP tests + bench sanity + DP self-test overhead
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
timeout 1200 python -m pytest tests/test_rccl_gpu.py tests/test_dp_gpu.py -x -q -m gpu > gpurun_out/r2_t1.log 2>&1; echo "rc=$?" >> gpurun_out/r2_t1.log
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "dropout_stream or load_state_dict or save_resume or reproducible" > gpurun_out/r2_t2.log 2>&1; echo "rc=$?" >> gpurun_out/r2_t2.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-variants --no-cpu-baseline > gpurun_out/r2_bench0.json 2> gpurun_out/r2_bench0.err
timeout 600 python bench.py --dp-selftest --steps 10 --warmup 3 > gpurun_out/r2_bench_dpself.json 2> gpurun_out/r2_bench_dpself.err
YTVLN_DP_GRAPH=single timeout 600 python bench.py --dp-selftest --steps 10 --warmup 3 > gpurun_out/r2_bench_dpself_single.json 2> gpurun_out/r2_bench_dpself_single.err
timeout 600 python bench.py --dp-selftest --graph off --steps 10 --warmup 3 > gpurun_out/r2_bench_dpself_eager.json 2> gpurun_out/r2_bench_dpself_eager.err
tail -5 gpurun_out/r2_t1.log gpurun_out/r2_t2.log; cat gpurun_out/r2_bench0.json | cut -c1-400; cut -c1-600 gpurun_out/r2_bench_dpself.json; cut -c1-300 gpurun_out/r2_bench_dpself_single.json; cut -c1-300 gpurun_out/r2_bench_dpself_eager.json; tail -3 gpurun_out/*.err
