"""torch.nn.functional.scaled_dot_product_attention (fp32, dropout 0.1) beside the ytvln attention kernels at the cfg-2 shapes: a yardstick."""
import os, sys, math
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "youtube-vln_amd"))
import torch
import torch.nn.functional as F
from ytvln import ops
dev = torch.device("cuda", 0)
N, p = 56, 0.1
st = ops.DropoutState(dev)


def timeit(f, iters=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1000


for name, h, d, Tq, Tk in [("img self", 8, 128, 288, 288), ("co t->v", 8, 128, 80, 288), ("co v->t", 8, 128, 288, 80), ("txt self", 12, 64, 80, 80)]:
    H = h * d
    q, k, v = (torch.randn(N * T, H, device=dev) for T in (Tq, Tk, Tk))
    mask = torch.zeros(N, Tk, device=dev)
    out, dout = torch.empty(N * Tq, H, device=dev), torch.randn(N * Tq, H, device=dev)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    sc = 1 / math.sqrt(d)
    lse = ops._attn_fwd(q, 0, H, k, 0, H, v, 0, H, mask, out, N, h, Tq, Tk, d, sc, p, st.tensor, 3)
    mine_f = timeit(lambda: ops._attn_fwd(q, 0, H, k, 0, H, v, 0, H, mask, out, N, h, Tq, Tk, d, sc, p, st.tensor, 3))
    mine_b = timeit(lambda: ops._attn_bwd(q, 0, H, k, 0, H, v, 0, H, mask, out, dout, lse, dq, 0, H, dk, 0, H, dv, 0, H, N, h, Tq, Tk, d, sc, p, st.tensor, 3))
    q4, k4, v4 = (t.view(N, -1, h, d).transpose(1, 2).detach().requires_grad_(True) for t in (q, k, v))
    m4 = mask.view(N, 1, 1, Tk)
    res = {}
    for label, ctx in (("default", None), ("math", "math")):
        try:
            if ctx == "math":
                from torch.nn.attention import sdpa_kernel, SDPBackend
                cm = sdpa_kernel([SDPBackend.MATH])
            else:
                import contextlib
                cm = contextlib.nullcontext()
            with cm:
                f = lambda: F.scaled_dot_product_attention(q4, k4, v4, attn_mask=m4, dropout_p=p)
                tf = timeit(f)
                o = f(); g = torch.randn_like(o)
                def fb():
                    o = F.scaled_dot_product_attention(q4, k4, v4, attn_mask=m4, dropout_p=p)
                    torch.autograd.grad(o, (q4, k4, v4), g)
                tfb = timeit(fb)
            res[label] = (tf, tfb - tf)
        except Exception as e:
            res[label] = (float("nan"), float("nan"))
    print(f"{name:9s} ytvln fwd {mine_f:7.1f} bwd {mine_b:7.1f} us | torch sdpa fwd {res['default'][0]:8.1f} bwd {res['default'][1]:8.1f} us | math fwd {res['math'][0]:8.1f} bwd {res['math'][1]:8.1f} us", flush=True)
