"""GPU box: 2 HIP training steps of the micro config, checkpoint written by THIS repo's save_model, then the third step.
Outputs gpurun_out/g9_repo_ckpt.bin + g9_repo_step3.npz for `python oracle/gen_golden_ckpt.py verify` (build container), which loads
the file into the REAL reference and checks that it continues identically."""
import json, os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "youtube-vln_amd"))
import numpy as np
import torch
from ytvln import synth, utils_init as U
from ytvln.lily import Lily
from ytvln.vilbert import BertConfig
from ytvln.vilbert_init import get_optimization

dev = torch.device("cuda", 0)
cfgd = json.load(open(os.path.join(ROOT, "youtube-vln_amd", "configs", "micro.json")))
cfgd.update(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, v_attention_probs_dropout_prob=0.0, v_hidden_dropout_prob=0.0)
args = types.SimpleNamespace(model_name="vilbert", ranking=True, traj_judge=True, masked_vision=True, masked_language=True, pretrain=True,
                             num_negatives=2, traj_loss_scale=1.0, not_traj_judge_data=False, local_rank=-1, skip_all_reduce=True,
                             weight_decay=0.01, learning_rate=1e-3, no_scheduler=False, ConstantLR=False, gradient_accumulation_steps=1,
                             num_epochs=1, warmup_proportion=0.2, cooldown_factor=2.0, resume=False)
cfg = BertConfig(**cfgd); cfg.args = args
model = Lily(cfg, dropout_prob=0.0)
shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(shapes, 11).items()})
model.to(dev).train()
batch = synth.to_torch(synth.make_batch(bs=2, K=3, T=8, frames=2, boxes=3, F=16, C=11, vocab=97, seed=21, ignore_rank_frac=0.0), dev)
opt, sched, _, _ = get_optimization(args, model, 10, None)
for i in range(2):
    U.train_step(model, opt, sched, batch, args, i, all_options=True)
out = os.path.join(ROOT, "gpurun_out"); os.makedirs(out, exist_ok=True)
U.save_model(out, "g9_repo_ckpt", None, model, opt, sched, epoch=4)
loss, _ = U.train_step(model, opt, sched, batch, args, 2, all_options=True)
res = {"loss3": np.array(float(loss))}
for n, p in model.named_parameters():
    res["p/" + n] = p.detach().cpu().numpy()
    st = opt.state.get(p, {})
    if "exp_avg" in st:
        res["m/" + n] = st["exp_avg"].cpu().numpy(); res["v/" + n] = st["exp_avg_sq"].cpu().numpy(); res["step/" + n] = np.array(st["step"])
np.savez_compressed(os.path.join(out, "g9_repo_step3.npz"), **res)
print("wrote", os.path.join(out, "g9_repo_ckpt.bin"), "loss3", float(loss))
