#!/bin/bash
# round-2 GPU pass 5: fused bias gradients + residual passthrough: full parity suite, bench, kernel trace, adoption diagnostics
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r2_full_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r2_full_tests.log
tail -15 gpurun_out/r2_full_tests.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-variants --no-cpu-baseline > gpurun_out/r2_bench1.json 2> gpurun_out/r2_bench1.err; cut -c1-300 gpurun_out/r2_bench1.json
YTVLN_DEBUG_ADOPT=1 timeout 300 python - > gpurun_out/r2_adopt.log 2>&1 <<'PYEOF'
import os, sys, json, types, collections
sys.path.insert(0, "youtube-vln_amd"); sys.argv = ["bench.py"]
import torch
import bench
from ytvln import synth, utils_init, optimization
from ytvln.lily import Lily
from ytvln.vilbert import BertConfig
from ytvln.vilbert_init import get_optimization
cfgname, bs, K, T, frames, boxes, flags = bench.WORKLOADS["cfg2_full_pretrain_bs8"]
args = bench.make_args(flags)
cfg = BertConfig.from_json_file(os.path.join("youtube-vln_amd", "configs", cfgname)); cfg.args = args
dev = torch.device("cuda", 0)
model = Lily(cfg).to(dev).train()
batch = synth.to_torch(synth.make_batch(bs=2, K=K, T=T, frames=frames, boxes=boxes, seed=1), dev)
opt, sched, _, _ = get_optimization(args, model, 10, None)
names = {tuple(p.shape): [] for p in model.parameters()}
for n, p in model.named_parameters(): names[tuple(p.shape)].append(n)
for i in range(3):
    optimization._DEBUG_ADOPT.clear()
    utils_init.train_step(model, opt, sched, batch, args, i, all_options=True)
c = collections.Counter(optimization._DEBUG_ADOPT)
print("re-adopted per step:", sum(c.values()))
for shape, k in c.most_common(): print(k, shape, names[shape][:3])
PYEOF
cat gpurun_out/r2_adopt.log | tail -30
TOPN=30 bash tools/kernel_stats.sh r2b > gpurun_out/r2b_kernel_stats.txt 2>&1
cat gpurun_out/r2b_kernel_stats.txt | cut -c1-160
