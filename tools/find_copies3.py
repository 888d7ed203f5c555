"""Who issues the device-to-device copies (__amd_rocclr_copyBuffer) of one eager training step?  torch.profiler with stacks around one step; every
aten::copy_ / clone / contiguous call that launched a Memcpy DtoD is attributed to its innermost ytvln (or torch.autograd) frame."""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "youtube-vln_amd")); sys.path.insert(0, ROOT)
import torch
from torch.profiler import ProfilerActivity, profile
import bench
from ytvln import ops, synth, utils_init
from ytvln.lily import Lily
from ytvln.vilbert import BertConfig
from ytvln.vilbert_init import get_optimization
dev = torch.device("cuda", 0)
ops.set_two_stream(False)
cfgname, bs, K, T, frames, boxes, flags = bench.WORKLOADS["cfg2_full_pretrain_bs8"]
args = bench.make_args(flags); args.local_rank = -1
cfg = BertConfig.from_json_file(os.path.join(ROOT, "youtube-vln_amd", "configs", cfgname)); cfg.args = args
torch.manual_seed(1234)
model = Lily(cfg).to(dev).train()
batch = synth.to_torch(synth.make_batch(bs=bs, K=K, T=T, frames=frames, boxes=boxes, seed=1234), dev)
opt, sched, _, _ = get_optimization(args, model, 20, None)
for s in range(3):
    utils_init.train_step(model, opt, sched, batch, args, s, all_options=True)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    utils_init.train_step(model, opt, sched, batch, args, 3, all_options=True)
    torch.cuda.synchronize()
evs = prof.events()
memcpy = [e for e in evs if "Memcpy" in e.name or "copyBuffer" in e.name]
print("device copy events:", len(memcpy), collections.Counter(e.name for e in memcpy).most_common(4))
cnt = collections.Counter()
for e in evs:
    if e.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::_to_copy", "aten::cat", "aten::add_", "aten::add") and e.device_type.name == "CPU":
        kids = [k for k in (e.cpu_children or []) if k.name.startswith("aten::")]
        if kids:
            continue        # count innermost aten ops only
        st = [f for f in (e.stack or []) if "/ytvln/" in f or "autograd" in f]
        where = st[0] if st else (e.stack[0] if e.stack else "?")
        shp = str(e.input_shapes[:1]) if e.input_shapes else ""
        cnt[(e.name, where[-90:], shp)] += 1
for (n, w, shp), c in cnt.most_common(40):
    print(f"{c:5d} {n:18s} {shp:28s} {w}")
