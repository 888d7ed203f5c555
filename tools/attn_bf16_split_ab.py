"""bf16 attention backward at head dimension 128 (BASELINE configs[4] shapes: 224 pairs, 8 heads): dK and dV in one pass with one wave per SIMD
(ATTN_DKV_SPLIT = 0) against two passes with two waves per SIMD (= 1); whole backward launch (delta + dQ + dK/dV) timed, interleaved, min of 3 passes;
the two forms must agree to bf16 rounding of the outputs."""
import math, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "youtube-vln_amd"))
import torch
from ytvln import _lib, ops
dev = torch.device("cuda", 0)
BF = torch.bfloat16
N, heads, d = int(os.environ.get("N", 224)), 8, 128
H = heads * d
scale = 1 / math.sqrt(d)
st = ops.DropoutState(dev)
heat = torch.randn(8192, 8192, device=dev).bfloat16()


def once(f, n=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1000


for name, Tq, Tk, p in (("image self-attention 576 x 576", 576, 576, 0.1), ("text over regions 80 x 576", 80, 576, 0.1), ("regions over text 576 x 80", 576, 80, 0.1),
                        ("image self-attention, no dropout", 576, 576, 0.0)):
    g = torch.Generator(device="cpu").manual_seed(Tq + Tk)
    q = (torch.randn(N * Tq, H, generator=g) * 0.5).to(dev).to(BF); k = (torch.randn(N * Tk, H, generator=g) * 0.5).to(dev).to(BF)
    v = (torch.randn(N * Tk, H, generator=g) * 0.5).to(dev).to(BF); dout = (torch.randn(N * Tq, H, generator=g) * 0.5).to(dev).to(BF)
    mask = torch.zeros(N, Tk, device=dev)
    out = torch.empty(N * Tq, H, device=dev, dtype=BF)
    lse = ops._attn_fwd(q, 0, H, k, 0, H, v, 0, H, mask, out, N, heads, Tq, Tk, d, scale, p, st.tensor if p else None, 3)
    res = {}
    best = {0: 1e30, 1: 1e30}
    outs = {}
    for ps in range(3):
        for _ in range(8):
            torch.matmul(heat, heat)
        for form in ((0, 1) if ps % 2 == 0 else (1, 0)):
            _lib.set_option("ATTN_DKV_SPLIT", form)
            dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
            f = lambda: ops._attn_bwd(q, 0, H, k, 0, H, v, 0, H, mask, out, dout, lse, dq, 0, H, dk, 0, H, dv, 0, H, N, heads, Tq, Tk, d, scale, p, st.tensor if p else None, 3)
            f(); torch.cuda.synchronize()
            best[form] = min(best[form], once(f))
            outs[form] = (dq.float(), dk.float(), dv.float())
    _lib.set_option("ATTN_DKV_SPLIT", 0)
    diffs = [float((a - b).abs().max() / a.abs().max()) for a, b in zip(outs[0], outs[1])]
    print(f"{name:36s} one pass {best[0]:8.1f} us   two passes {best[1]:8.1f} us   ({best[0] / best[1]:.2f}x)   max rel diff dq/dk/dv {diffs[0]:.1e} {diffs[1]:.1e} {diffs[2]:.1e}", flush=True)
