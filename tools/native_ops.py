"""Which torch (non-ytvln) device launches remain on one training step, and from which line of the host code: torch.profiler with stacks over one
eager cfg-2 step.   usage: python tools/native_ops.py [precision]"""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "youtube-vln_amd")); sys.path.insert(0, ROOT)
import torch
import bench
from torch.profiler import profile, ProfilerActivity

sys.argv = [sys.argv[0]] + ["--no-cpu-baseline", "--no-variants", "--no-kernel-timing", "--steps", "1", "--warmup", "1", "--graph", "off", "--host-probe", "0"] + \
    (["--precision", sys.argv[1]] if len(sys.argv) > 1 else [])
orig = None
from ytvln import utils_init
calls = {"n": 0}
real = utils_init.train_step
def wrapped(*a, **k):
    calls["n"] += 1
    if calls["n"] == 2:
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
            out = real(*a, **k)
            torch.cuda.synchronize()
        agg = collections.Counter(); tim = collections.Counter()
        for ev in prof.events():
            if ev.device_type == torch.autograd.DeviceType.CPU and ev.name.startswith("aten::") and ev.name.split("::")[1] in (
                    "copy_", "fill_", "zero_", "cat", "add", "add_", "mul", "mul_", "sum", "to", "_to_copy", "clone", "contiguous", "div", "sub", "rsub", "index", "masked_fill", "eq", "ne", "where"):
                stack = [s for s in ev.stack if "ytvln" in s or "bench.py" in s]
                key = (ev.name, stack[0].strip()[-90:] if stack else "?")
                agg[key] += 1
                tim[key] += ev.device_time_total if hasattr(ev, "device_time_total") else 0
        for (name, where), n in agg.most_common(60):
            print(f"{n:5d}  {name:18s} {tim[(name, where)]:9.1f} us  {where}", file=sys.stderr)
        return out
    return real(*a, **k)
utils_init.train_step = wrapped
bench.main()
