"""Run-to-run reproducibility of the bf16-resident kernels under concurrency: every launch is deterministic by construction (fixed-order
reductions), so the same launch repeated while a second HIP stream keeps the chip busy must give the same bits.  A mismatch = a race inside
the kernel (a missing wait / barrier) that co-running workgroups expose.   usage: python tools/bf16_repro.py [reps]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "youtube-vln_amd"))
import torch
from ytvln import ops

dev = torch.device("cuda", 0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
BF = torch.bfloat16
g = torch.Generator(device=dev); g.manual_seed(1)
rn = lambda *s: (torch.randn(*s, device=dev, generator=g) * 0.5).to(BF)
side = torch.cuda.Stream()
bg_a, bg_b, bg_c = rn(8192, 1024), rn(1024, 1024), torch.empty(8192, 1024, dtype=BF, device=dev)

def background(n):
    with torch.cuda.stream(side):
        for _ in range(n):
            ops._gemm_bf16(bg_a, 1024, 0, bg_b, 1024, 1, bg_c, 1024, 8192, 1024, 1024)

def check(name, launch, outs):
    launch(); torch.cuda.synchronize()
    ref = [o.clone() for o in outs]
    bad_alone = bad_conc = 0
    for i in range(reps):
        for o in outs: o.zero_()
        launch(); torch.cuda.synchronize()
        bad_alone += any(not torch.equal(o, r) for o, r in zip(outs, ref))
    for i in range(reps):
        for o in outs: o.zero_()
        torch.cuda.synchronize()
        background(6); launch(); background(6); torch.cuda.synchronize()
        if any(not torch.equal(o, r) for o, r in zip(outs, ref)):
            bad_conc += 1
            if bad_conc == 1:
                d = [(o.float() - r.float()).abs() for o, r in zip(outs, ref)]
                print(f"   first mismatch: max|d| {[float(x.max()) for x in d]}, elements {[int((x > 0).sum()) for x in d]} of {[x.numel() for x in d]}")
    print(f"{name:58s} alone {bad_alone}/{reps}  concurrent {bad_conc}/{reps}", flush=True)

rows = {"text cfg4": 96 * 80, "pooled": 96} if os.environ.get("GEMM_SHORT") else {"text cfg4": 96 * 80, "image cfg4": 96 * 252, "pooled": 96, "text cfg2": 56 * 80}
for tag, M in rows.items():
    for N, K in ((768, 768), (2304, 768), (3072, 768), (768, 3072), (1024, 1024), (1024, 768)):
        if tag == "pooled" and (N, K) not in ((1024, 768), (1024, 1024)):
            continue
        x, w, dy = rn(M, K), rn(N, K), rn(M, N)
        y = torch.empty(M, N, dtype=BF, device=dev)
        check(f"{tag} fwd  y[{M},{N}] = x W^T  K={K}", lambda: ops._gemm_bf16(x, K, 0, w, K, 1, y, N, M, N, K), [y])
        dx = torch.empty(M, K, dtype=BF, device=dev)
        check(f"{tag} dX   [{M},{K}] = dY W    N={N}", lambda: ops._gemm_bf16(dy, N, 0, w, K, 0, dx, K, M, K, N), [dx])
        dw = torch.empty(N, K, dtype=torch.float32, device=dev); db = torch.empty(N, dtype=torch.float32, device=dev)
        check(f"{tag} dW   [{N},{K}] = dY^T X  M={M} (+rowsum)", lambda: ops._gemm_bf16(dy, N, 1, x, K, 0, dw, K, N, K, M, rowsum=db), [dw, db])

# ---- attention (bf16-resident kernels), forward + backward, p = 0 -----------------------------------------------------------------------
def attn_case(name, N, T, heads, d):
    H = heads * d
    qkv = rn(N * T, 3 * H).requires_grad_()
    mask = torch.zeros(N, T, device=dev); mask[:, T - 3:] = -10000.0
    do = rn(N * T, H)
    res = {}
    def launch():
        qkv.grad = None
        out, lse = ops.SelfAttentionFn.apply(qkv, mask, N, T, heads, 0.0, None, 0)
        out.backward(do)
        res["o"], res["g"] = out.detach(), qkv.grad
    launch(); torch.cuda.synchronize()
    ref = (res["o"].clone(), res["g"].clone())
    bad = [0, 0]
    for conc in (0, 1):
        for i in range(reps):
            torch.cuda.synchronize()
            if conc: background(6)
            launch()
            if conc: background(6)
            torch.cuda.synchronize()
            bad[conc] += (not torch.equal(res["o"], ref[0])) or (not torch.equal(res["g"], ref[1]))
    print(f"{name:58s} alone {bad[0]}/{reps}  concurrent {bad[1]}/{reps}", flush=True)

def co_case(name, N, T, R, heads, d):
    H = heads * d
    q1, kv1 = rn(N * R, H).requires_grad_(), rn(N * R, 2 * H).requires_grad_()
    q2, kv2 = rn(N * T, H).requires_grad_(), rn(N * T, 2 * H).requires_grad_()
    m1 = torch.zeros(N, R, device=dev); m2 = torch.zeros(N, T, device=dev); m2[:, T - 3:] = -10000.0
    d1, d2 = rn(N * T, H), rn(N * R, H)
    res = {}
    leaves = (q1, kv1, q2, kv2)
    def launch():
        for l in leaves: l.grad = None
        c1, c2, l1, l2 = ops.CoAttentionFn.apply(q1, kv1, q2, kv2, m1, m2, N, R, T, heads, 0.0, 0.0, None, 0, 0)
        torch.autograd.backward([c1, c2], [d1, d2])
        res["v"] = [c1.detach(), c2.detach()] + [l.grad for l in leaves]
    launch(); torch.cuda.synchronize()
    ref = [v.clone() for v in res["v"]]
    bad = [0, 0]
    for conc in (0, 1):
        for i in range(reps):
            torch.cuda.synchronize()
            if conc: background(6)
            launch()
            if conc: background(6)
            torch.cuda.synchronize()
            bad[conc] += any(not torch.equal(a, b) for a, b in zip(res["v"], ref))
    print(f"{name:58s} alone {bad[0]}/{reps}  concurrent {bad[1]}/{reps}", flush=True)

attn_case("self-attention text  N=96 T=80 h=12 d=64", 96, 80, 12, 64)
attn_case("self-attention image N=96 R=252 h=8 d=128", 96, 252, 8, 128)
attn_case("self-attention image N=56 R=288 h=8 d=128", 56, 288, 8, 128)
co_case("co-attention N=96 T=80 R=252 h=8 d=128", 96, 80, 252, 8, 128)
co_case("co-attention N=56 T=80 R=288 h=8 d=128", 56, 80, 288, 8, 128)

# ---- LayerNorm (bf16 rows), forward + backward -------------------------------------------------------------------------------------------
def ln_case(name, rows, H):
    x, r = rn(rows, H).requires_grad_(), rn(rows, H).requires_grad_()
    w = torch.randn(H, device=dev).requires_grad_(); b = torch.randn(H, device=dev).requires_grad_()
    dy = rn(rows, H)
    res = {}
    def launch():
        for l in (x, r, w, b): l.grad = None
        y = ops.add_layer_norm(x, r, w, b, 1e-12, 0.0, 0.0, None)
        y.backward(dy)
        res["v"] = [y.detach(), x.grad, r.grad, w.grad, b.grad]
    launch(); torch.cuda.synchronize()
    ref = [v.clone() for v in res["v"]]
    bad = [0, 0]
    for conc in (0, 1):
        for i in range(reps):
            torch.cuda.synchronize()
            if conc: background(6)
            launch()
            if conc: background(6)
            torch.cuda.synchronize()
            bad[conc] += any(not torch.equal(a, b) for a, b in zip(res["v"], ref))
    print(f"{name:58s} alone {bad[0]}/{reps}  concurrent {bad[1]}/{reps}", flush=True)

ln_case("add+LayerNorm text rows 7680 x 768", 7680, 768)
ln_case("add+LayerNorm image rows 24192 x 1024", 24192, 1024)
