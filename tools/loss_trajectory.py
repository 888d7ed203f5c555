"""Loss trajectories of the full model on ONE synthetic cfg-2 batch (dropout on, same mask stream) for the three projection arithmetics:
native fp32 MFMA, fp32x3 and bf16.  Evidence that fp32x3 trains like fp32 (LABNOTES.md 5a) and that the step is numerically healthy over
hundreds of optimizer steps.  usage: python tools/loss_trajectory.py [steps] [tag] -> gpurun_out/<tag>_loss_trajectory.json (tag: round2)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "youtube-vln_amd")); sys.path.insert(0, ROOT)
import torch
import bench
from ytvln import ops, synth, utils_init
from ytvln.lily import Lily
from ytvln.vilbert import BertConfig
from ytvln.vilbert_init import get_optimization

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
tag = sys.argv[2] if len(sys.argv) > 2 else "round2"
cfgname, bs, K, T, frames, boxes, flags = bench.WORKLOADS["cfg2_full_pretrain_bs8"]
dev = torch.device("cuda", 0)
out = {"workload": "cfg2_full_pretrain_bs8 (56 pairs), one fixed synthetic batch, lr 4e-5 WarmupLinear, dropout on, seed 1234", "steps": steps, "loss": {}}
for prec in ("fp32", "fp32x3", "bf16"):
    torch.manual_seed(1234)
    ops.DropoutState.manual_seed(1234)
    ops.set_matmul_precision(prec)
    args = bench.make_args(flags)
    cfg = BertConfig.from_json_file(os.path.join(ROOT, "youtube-vln_amd", "configs", cfgname)); cfg.args = args
    model = Lily(cfg).to(dev).train()
    batch = synth.to_torch(synth.make_batch(bs=bs, K=K, T=T, frames=frames, boxes=boxes, seed=1234), dev)
    opt, sched, _, _ = get_optimization(args, model, steps, None)
    curve = []
    t0 = time.time()
    for i in range(steps):
        loss, _ = utils_init.train_step(model, opt, sched, batch, args, i, all_options=True)
        if i % 10 == 0 or i == steps - 1:
            curve.append((i, round(float(loss), 5)))
    out["loss"][prec] = curve
    print(prec, curve[0], curve[len(curve) // 2], curve[-1], f"{time.time() - t0:.0f} s", flush=True)
    del model, opt
    torch.cuda.empty_cache()
ops.set_matmul_precision("fp32")
a, b = dict(out["loss"]["fp32"]), dict(out["loss"]["fp32x3"])
out["max_abs_diff_fp32x3_vs_fp32"] = max(abs(a[k] - b[k]) for k in a)
c = dict(out["loss"]["bf16"])
out["max_rel_diff_bf16_vs_fp32"] = max(abs(a[k] - c[k]) / max(abs(a[k]), 1e-6) for k in a)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"{tag}_loss_trajectory.json"), "w"), indent=1)
print("max |fp32x3 - fp32| over the trajectory:", out["max_abs_diff_fp32x3_vs_fp32"], " max rel |bf16 - fp32|:", out["max_rel_diff_bf16_vs_fp32"])
