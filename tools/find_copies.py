"""Where do the device-to-device copies / fills / small elementwise launches of one training step come from?  Three eager steps of the
headline workload (one stream) under torch.profiler with Python stacks; prints, per (kernel name, innermost ytvln frame), launches per step."""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "youtube-vln_amd")); sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
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
STEPS = 2
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    for s in range(STEPS):
        utils_init.train_step(model, opt, sched, batch, args, 3 + s, all_options=True)
    torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    n = ev.name
    if not (n.startswith("aten::copy_") or n.startswith("aten::fill_") or n.startswith("aten::zero_") or n.startswith("aten::add") or n.startswith("aten::mul")
            or n.startswith("aten::cat") or n.startswith("aten::clone") or n.startswith("aten::contiguous") or n.startswith("aten::sum") or n.startswith("aten::_to_copy")):
        continue
    if ev.device_type != torch.autograd.DeviceType.CPU:
        continue
    st = [f for f in (ev.stack or []) if "ytvln" in f or "bench.py" in f]
    where = st[0].strip()[-80:] if st else ""
    shapes = str([tuple(x) if isinstance(x, (list, tuple)) else x for x in (ev.input_shapes or [])])[:90]
    cnt[(n, shapes + " " + where)] += 1
for (n, w), c in sorted(cnt.items(), key=lambda kv: -kv[1])[:70]:
    print(f"{c / STEPS:7.1f}/step  {n:22s} {w}")
