"""Attention micro-benchmark at the cfg-2 shapes (N = 56 pairs per GPU = global bs 64 on 8 GPUs).  CASES=co FWD_ONLY=1 isolates the
BertBiAttention forward (fused QK^T -> softmax -> dropout -> PV kernel) for the north-star MFMA-utilisation counter run
(tools/coattn_pmc.sh)."""
import os, sys, math
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "youtube-vln_amd"))
import torch
from ytvln import _lib
if os.environ.get("YTVLN_LIB"):          # experiment builds (e.g. scratch/lib_probe.so: timing probes of the attention forward)
    _lib.LIB_PATH = os.path.abspath(os.environ["YTVLN_LIB"])
from ytvln import ops
dev = torch.device("cuda", 0)
N = int(os.environ.get("PAIRS_N", "56"))            # PAIRS_N=224 REGIONS=576: the cfg-5 shapes
RG = int(os.environ.get("REGIONS", "288"))
TX = int(os.environ.get("TEXT", "80"))             # TEXT=64|96: the 32-row neighbours of the 80-token text (what a 16-row tail could save at most)
cases = [("img self", 8, 128, RG, RG), ("co t->v", 8, 128, TX, RG), ("co v->t", 8, 128, RG, TX), ("txt self", 12, 64, TX, TX)]
only = os.environ.get("CASES")          # e.g. CASES="co" -> only the BertBiAttention shapes
if only:
    cases = [c for c in cases if c[0].startswith(only)]
fwd_only = bool(os.environ.get("FWD_ONLY"))
ops.set_matmul_precision(os.environ.get("PRECISION", "fp32"))      # PRECISION=bf16 -> the bf16-resident kernels (bf16 tensors in HBM)
DT = torch.bfloat16 if os.environ.get("PRECISION") == "bf16" else torch.float32
p = float(os.environ.get("PDROP", "0.1"))
st = ops.DropoutState(dev)
for name, h, d, Tq, Tk in cases:
    H = h * d
    q, k, v = (torch.randn(N * T, H, device=dev).to(DT) for T in (Tq, Tk, Tk))
    mask = torch.zeros(N, Tk, device=dev)
    out, dout = torch.empty(N * Tq, H, device=dev, dtype=DT), torch.randn(N * Tq, H, device=dev).to(DT)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    sc = 1 / math.sqrt(d)
    def fwd(): return ops._attn_fwd(q, 0, H, k, 0, H, v, 0, H, mask, out, N, h, Tq, Tk, d, sc, p, st.tensor, 3)
    lse = fwd()
    def bwd(): ops._attn_bwd(q, 0, H, k, 0, H, v, 0, H, mask, out, dout, lse, dq, 0, H, dk, 0, H, dv, 0, H, N, h, Tq, Tk, d, sc, p, st.tensor, 3)
    res = []
    for f, mult in (((fwd, 4.0),) if fwd_only else ((fwd, 4.0), (bwd, 14.0))):   # fwd 2 matmuls, bwd 7 (incl. recompute) -> 2*Tq*Tk*d each
        for _ in range(3): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        res.append((ms, mult * N * h * Tq * Tk * d / ms / 1e9))
    line = f"{name:9s} fwd {res[0][0]*1000:7.1f} us {res[0][1]:6.1f} TF/s"
    if not fwd_only:
        line += f" | bwd {res[1][0]*1000:7.1f} us {res[1][1]:6.1f} TF/s"
    print(line + " (algorithmic flops 4*N*h*Tq*Tk*d fwd, 14*... bwd)", flush=True)

# both BertBiAttention directions in ONE launch per kernel (ytvln_attn_fwd_pair / ytvln_attn_bwd_pair), as CoAttentionFn runs them
if not only or only.startswith("co"):
    h, d, T, R = 8, 128, TX, RG
    Hb = h * d
    q1, kv1 = torch.randn(N * R, Hb, device=dev).to(DT), torch.randn(N * R, 2 * Hb, device=dev).to(DT)
    q2, kv2 = torch.randn(N * T, Hb, device=dev).to(DT), torch.randn(N * T, 2 * Hb, device=dev).to(DT)
    m1, m2 = torch.zeros(N, R, device=dev), torch.zeros(N, T, device=dev)
    for t in (q1, kv1, q2, kv2):
        t.requires_grad_(True)
    st2 = ops.DropoutState(dev)

    def pair_fwd():
        return ops.CoAttentionFn.apply(q1, kv1, q2, kv2, m1, m2, N, R, T, h, p, p, st2.tensor if p > 0 else None, 5, 6)
    c1, c2, _, _ = pair_fwd()
    g1, g2 = torch.randn_like(c1), torch.randn_like(c2)

    def pair_fb():
        a, b, _, _ = pair_fwd()
        torch.autograd.backward([a, b], [g1, g2])
    res = []
    for f in (pair_fwd,) if fwd_only else (pair_fwd, pair_fb):
        for _ in range(3): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 10 * 1000)
    fl = 2 * 4.0 * N * h * T * R * d
    line = f"co pair   fwd {res[0]:7.1f} us {fl / res[0] / 1e6:6.1f} TF/s"
    if not fwd_only:
        line += f" | fwd+bwd {res[1]:7.1f} us (bwd {res[1] - res[0]:7.1f} us {3.5 * fl / (res[1] - res[0]) / 1e6:6.1f} TF/s)"
    print(line + "   <- both directions, one launch per kernel", flush=True)
