"""Which lines of ytvln/ issue the small torch launches of one eager training step (copy_ / fill_ / zero_ / clone / contiguous / to / float / mul / add /
sum ...)?  The Tensor methods and torch functions are wrapped in Python and every call records its innermost ytvln frame (the autograd
thread included); prints calls per step."""
import collections, os, sys, traceback
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
cnt = collections.Counter()
ON = [False]


def where():
    for f in reversed(traceback.extract_stack()[:-2]):
        if "/ytvln/" in f.filename:
            return f"{os.path.basename(f.filename)}:{f.lineno} {f.line[:70]}"
    return "?"


def wrap(owner, name):
    orig = getattr(owner, name)

    def w(*a, **k):
        if ON[0]:
            t = a[0] if a and isinstance(a[0], torch.Tensor) else None
            if t is None or t.is_cuda:
                cnt[(name, where(), tuple(t.shape) if t is not None else ())] += 1
        return orig(*a, **k)
    setattr(owner, name, w)


for m in ["copy_", "fill_", "zero_", "clone", "contiguous", "to", "float", "double", "long", "int", "bfloat16", "sum", "mul", "mul_", "add", "add_", "sub", "div",
          "__mul__", "__add__", "__sub__", "__truediv__", "__rmul__", "__radd__", "__iadd__", "__imul__", "__setitem__", "__getitem__", "masked_fill", "masked_fill_",
          "new_zeros", "new_full", "new_ones", "mean", "item", "index_select", "gather", "scatter_", "eq", "ne", "__eq__", "__ne__", "__ge__", "__gt__", "argmax", "max"]:
    wrap(torch.Tensor, m)
for m in ["zeros", "ones", "full", "cat", "stack", "zeros_like", "ones_like", "full_like", "where", "arange", "tensor", "sum"]:
    wrap(torch, m)
STEPS = 2
ON[0] = True
for s in range(STEPS):
    utils_init.train_step(model, opt, sched, batch, args, 3 + s, all_options=True)
torch.cuda.synchronize()
ON[0] = False
for (n, w, sh), c in sorted(cnt.items(), key=lambda kv: -kv[1])[:90]:
    print(f"{c / STEPS:6.1f}/step {n:12s} {str(sh):18s} {w}")
