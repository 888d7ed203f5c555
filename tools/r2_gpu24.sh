#!/bin/bash
# round-2 GPU pass 24: bias gradient riding on the bf16 staging pass -- kernel test, bf16 model parity tests, cfg5 bench A/B
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "cast_bf16 or bf16" 2>&1 | tail -2
timeout 1500 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "bf16" 2>&1 | tail -2
YTVLN_FUSED_BIAS_GRAD=0 timeout 900 python bench.py --workload cfg5_long_traj_bs32 --precision bf16 --steps 6 --warmup 2 --no-cpu-baseline --no-variants > gpurun_out/r2_cfg5_unfused.json 2> gpurun_out/r2_cfg5_unfused.err
timeout 900 python bench.py --workload cfg5_long_traj_bs32 --precision bf16 --steps 6 --warmup 2 --no-cpu-baseline --no-variants > gpurun_out/r2_cfg5_fused.json 2> gpurun_out/r2_cfg5_fused.err
python - <<'PY'
import json
for n in ("unfused", "fused"):
    try:
        d = json.loads([l for l in open(f"gpurun_out/r2_cfg5_{n}.json") if l.startswith("{")][0]); print(n, d["value"], d["ms_per_step"], d["final_loss"])
    except Exception as e:
        print(n, "failed", e, open(f"gpurun_out/r2_cfg5_{n}.err").read()[-800:])
PY
