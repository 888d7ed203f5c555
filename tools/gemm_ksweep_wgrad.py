import os, sys
sys.path.insert(0, "/root/repo/youtube-vln_amd")
import numpy as np, torch
from ytvln import ops
dev = torch.device("cuda", 0)
def run(M, N, K, ta, tb, iters=20):
    A = torch.randn((K, M) if ta else (M, K), device=dev); B = torch.randn((N, K) if tb else (K, N), device=dev); C = torch.empty(M, N, device=dev)
    lda = M if ta else K; ldb = K if tb else N
    for _ in range(3): ops._gemm(A, lda, ta, B, ldb, tb, C, N, M, N, K)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): ops._gemm(A, lda, ta, B, ldb, tb, C, N, M, N, K)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1000
for (M, N) in [(1024, 1024), (2048, 1024), (3072, 1024), (768, 3072)]:
    Ks = [4096, 8192, 16384, 32768]
    us = [run(M, N, K, 1, 0) for K in Ks]
    per_k, fixed = np.polyfit(Ks, us, 1)
    ideal = 2.0 * M * N / 157.3e6
    print(f"{M:5d} {N:5d} tA1 tB0 " + " ".join(f"K={k}:{u:7.1f}" for k, u in zip(Ks, us)) + f"  fixed {fixed:6.1f} us per-k {per_k*1000:6.2f} ns (peak {ideal*1000:6.2f}) main loop {ideal/per_k:5.3f}", flush=True)
