"""K sweep of one output shape per operand-layout variant: time = fixed + per_k * K separates the per-launch cost (ramp, prologue,
epilogue store burst, drain) from the main-loop rate.  HIP events, 20 launches each."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "youtube-vln_amd"))
import numpy as np
import torch
from ytvln import ops
dev = torch.device("cuda", 0)
if os.environ.get("PRECISION"):
    ops.set_matmul_precision(os.environ["PRECISION"])


def run(M, N, K, ta, tb, iters=20):
    A = torch.randn((K, M) if ta else (M, K), device=dev)
    B = torch.randn((N, K) if tb else (K, N), device=dev)
    C = torch.empty(M, N, device=dev)
    lda = M if ta else K; ldb = K if tb else N
    for _ in range(3): ops._gemm(A, lda, ta, B, ldb, tb, C, N, M, N, K)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): ops._gemm(A, lda, ta, B, ldb, tb, C, N, M, N, K)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1000


for (M, N) in [(16128, 1024), (16128, 3072), (4480, 768), (4480, 3072)]:
    for ta, tb in [(0, 1), (0, 0)]:
        Ks = [256, 512, 1024, 2048, 4096]
        us = [run(M, N, K, ta, tb) for K in Ks]
        per_k, fixed = np.polyfit(Ks[1:], us[1:], 1)
        ideal = 2.0 * M * N / 157.3e6        # us per k at the fp32 MFMA peak
        print(f"{M:6d} {N:5d} tA{ta} tB{tb}  " + " ".join(f"K={k}:{u:7.1f}" for k, u in zip(Ks, us)) +
              f"   fixed {fixed:6.1f} us  per-k {per_k * 1000:6.1f} ns (peak {ideal * 1000:6.1f} ns -> main loop {ideal / per_k:5.3f} of peak)", flush=True)
