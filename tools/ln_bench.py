"""LayerNorm backward (ytvln_ln_bwd_f32) alone at the cfg-2 shapes, HIP events, behind a warm-up: us per launch and algorithmic TB/s
(reads dy, s; writes ds and -- with dropout in front of the residual add -- dx)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "youtube-vln_amd"))
import torch
from ytvln import ops
dev = torch.device("cuda", 0)
rng = torch.tensor([1234, 0], dtype=torch.int64, device=dev)
heat = torch.randn(4096, 4096, device=dev)
for rows, H in [(16128, 1024), (4480, 768), (4480, 1024), (16128, 768)]:
    for p_pre in (0.1, 0.0):
        dy = torch.randn(rows, H, device=dev); s = torch.randn(rows, H, device=dev)
        mean = s.mean(1); rstd = 1.0 / s.std(1)
        gamma = torch.randn(H, device=dev)
        for _ in range(10):
            heat @ heat
        best = 1e9
        for rep in range(3):
            for _ in range(5):
                ops._ln_bwd(dy, s, mean, rstd, gamma, rows, H, p_pre, 0.0, rng, 3)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops._ln_bwd(dy, s, mean, rstd, gamma, rows, H, p_pre, 0.0, rng, 3)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 20 * 1000)
        nbytes = rows * H * 4 * (4 if p_pre > 0 else 3)
        print(f"ln_bwd+colsum rows {rows:6d} H {H:5d} p_pre {p_pre:.1f}: {best:7.1f} us  {nbytes / best / 1e6:5.2f} TB/s (incl. the two colsum stages)", flush=True)

# forward: y = LN(dropout(x) + res), saving the pre-norm sum and the row statistics (reads x, res; writes y, s)
from ytvln.ops import call, _ptr, _stream
for rows, H in [(16128, 1024), (4480, 768), (4480, 1024), (16128, 768)]:
    x = torch.randn(rows, H, device=dev); res = torch.randn(rows, H, device=dev)
    gamma = torch.randn(H, device=dev); beta = torch.randn(H, device=dev)
    y = torch.empty_like(x); s = torch.empty_like(x)
    mean = torch.empty(rows, device=dev); rstd = torch.empty(rows, device=dev)
    f = lambda: call("ytvln_ln_fwd_f32", _ptr(x), _ptr(res), _ptr(gamma), _ptr(beta), _ptr(y), _ptr(s), _ptr(mean), _ptr(rstd), rows, H,
                     1e-12, 0.1, 0.0, _ptr(rng), 3, _stream())
    for _ in range(10):
        heat @ heat
    best = 1e9
    for rep in range(3):
        for _ in range(5):
            f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20 * 1000)
    print(f"ln_fwd rows {rows:6d} H {H:5d} p_pre 0.1: {best:7.1f} us  {rows * H * 16 / best / 1e6:5.2f} TB/s", flush=True)
