"""Soak run of the captured training step: the full cfg-2 model, hipGraph replay, a DIFFERENT synthetic batch copied into the static input
tensors before every replay (8 batches in rotation), AdamW + WarmupLinear, dropout on.  Checks every 25 steps that the loss is finite,
that device memory does not grow, and reports the loss curve (random labels: the loss settles at the chance level of the four heads).
usage: python tools/soak.py [steps] [tag] [precision] [workload] -> gpurun_out/<tag>_soak.json (tag: round2, precision: fp32, workload: cfg2_full_pretrain_bs8)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "youtube-vln_amd")); sys.path.insert(0, ROOT)
import torch
import bench
from ytvln import ops, synth, utils_init
from ytvln.lily import Lily
from ytvln.vilbert import BertConfig
from ytvln.vilbert_init import get_optimization

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
tag = sys.argv[2] if len(sys.argv) > 2 else "round2"
precision = sys.argv[3] if len(sys.argv) > 3 else "fp32"
workload = sys.argv[4] if len(sys.argv) > 4 else "cfg2_full_pretrain_bs8"
cfgname, bs, K, T, frames, boxes, flags = bench.WORKLOADS[workload]
dev = torch.device("cuda", 0)
torch.manual_seed(1234)
ops.DropoutState.manual_seed(1234)
ops.set_matmul_precision(precision)
args = bench.make_args(flags)
cfg = BertConfig.from_json_file(os.path.join(ROOT, "youtube-vln_amd", "configs", cfgname)); cfg.args = args
model = Lily(cfg).to(dev).train()
pool = [synth.to_torch(synth.make_batch(bs=bs, K=K, T=T, frames=frames, boxes=boxes, seed=2000 + i), dev) for i in range(8)]
static = [t.clone() if torch.is_tensor(t) else t for t in pool[0]]
opt, sched, _, _ = get_optimization(args, model, steps, None)
for i in range(2):
    utils_init.train_step(model, opt, sched, static, args, i, all_options=True)
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    loss, _ = utils_init.train_step(model, opt, None, static, args, 0, all_options=True)
torch.cuda.synchronize()
curve, mem = [], []
t0 = time.time()
for i in range(steps):
    for dst, src in zip(static, pool[i % 8]):
        if torch.is_tensor(dst):
            dst.copy_(src)
    opt.prepare_replay()
    graph.replay()
    sched.step()
    if i % 25 == 0 or i == steps - 1:
        v = float(loss)
        assert v == v and abs(v) < 1e4, (i, v)
        curve.append((i, round(v, 4)))
        mem.append(round(torch.cuda.memory_reserved(dev) / 2 ** 30, 3))
torch.cuda.synchronize()
el = time.time() - t0
bad = [n for n, p in model.named_parameters() if not torch.isfinite(p).all()]
out = {"workload": f"{workload} ({bs * K} pairs/step), precision {precision}, hipGraph replay, 8 synthetic batches in rotation copied into the static inputs, dropout on",
       "steps": steps, "seconds": round(el, 1), "pairs_per_s_including_refill": round(steps * bs * K / el, 1), "loss": curve,
       "memory_reserved_gb": {"first": mem[0], "last": mem[-1], "max": max(mem)}, "non_finite_parameters": bad}
assert not bad and mem[-1] == mem[0], out
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"{tag}_soak.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "loss"}), curve[0], curve[len(curve) // 2], curve[-1])
