"""Which parameters' gradients land OUTSIDE the optimizer's flat gradient arena after a backward pass (each costs a copy in
FusedAdamW._ensure_arena)?  One eager step of the headline workload; prints the parameter names."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "youtube-vln_amd")); sys.path.insert(0, ROOT)
import torch
import bench
from ytvln import ops, synth, utils_init
from ytvln.lily import Lily
from ytvln.vilbert import BertConfig
from ytvln.vilbert_init import get_optimization
dev = torch.device("cuda", 0)
ops.set_two_stream(False)
cfgname, bs, K, T, frames, boxes, flags = bench.WORKLOADS["cfg2_full_pretrain_bs8"]
args = bench.make_args(flags)
args.local_rank = -1
cfg = BertConfig.from_json_file(os.path.join(ROOT, "youtube-vln_amd", "configs", cfgname))
cfg.args = args
torch.manual_seed(1234)
model = Lily(cfg).to(dev).train()
batch = synth.to_torch(synth.make_batch(bs=bs, K=K, T=T, frames=frames, boxes=boxes, seed=1234), dev)
opt, sched, _, _ = get_optimization(args, model, 20, None)
for s in range(3):
    utils_init.train_step(model, opt, sched, batch, args, s, all_options=True)
torch.cuda.synchronize()
utils_init.train_step(model, opt, None, batch, args, 3, all_options=True, optimizer_step=False)
torch.cuda.synchronize()
ar = opt._arena
lo, hi = ar["g"].data_ptr(), ar["g"].data_ptr() + 4 * ar["g"].numel()
n = 0
for name, p in model.named_parameters():
    if p.grad is None:
        continue
    if not (lo <= p.grad.data_ptr() < hi):
        n += 1
        print("stray", name, tuple(p.shape))
print("total stray:", n)
