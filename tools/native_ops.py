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
        rows = []
        for ev in prof.key_averages(group_by_stack_n=8):
            if ev.key.startswith("aten::") and ev.key.split("::")[1] in (
                    "copy_", "fill_", "zero_", "cat", "add", "add_", "mul", "mul_", "sum", "div", "sub", "rsub", "eq", "masked_fill", "where", "index", "neg", "clamp"):
                stack = [s_ for s_ in ev.stack if ("ytvln" in s_ or "bench.py" in s_) and "native_ops" not in s_]
                rows.append((ev.count, ev.key, ev.device_time_total, " <- ".join(x.strip().split("/")[-1][:70] for x in stack[:3]) or "?"))
        rows.sort(key=lambda r: -r[0])
        for n, name, t, where in rows[:50]:
            print(f"{n:5d}  {name:14s} {t:9.1f} us  {where}", file=sys.stderr)
        return out
    return real(*a, **k)
utils_init.train_step = wrapped
bench.main()
