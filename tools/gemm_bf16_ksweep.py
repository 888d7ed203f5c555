"""bf16 GEMM, K sweep at the cfg-5 image shape (M = 129024, N = 1024): time = fixed + per_k * K per launch separates what a tile pays once
(dispatch, first operands, the epilogue's store burst) from the main-loop rate; the vendor library (torch.matmul) on the same operands.
Run once per form: YTVLN_GEMM_TILE=0|4, YTVLN_GEMM_BF16_PERSIST=0|1."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "youtube-vln_amd"))
import numpy as np
import torch
from ytvln import ops
dev = torch.device("cuda", 0)
M = int(os.environ.get("M", 129024)); N = int(os.environ.get("N", 1024))
Ks = [64, 128, 256, 512, 1024, 2048]


def timed(f, n=10):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1000


heat_a = torch.randn(8192, 8192, device=dev).bfloat16()
for tb in (1, 0):
    for cdt in (torch.bfloat16, torch.float32):
        mine, lib = [], []
        for K in Ks:
            A = torch.randn(M, K, device=dev).bfloat16()
            B = torch.randn((N, K) if tb else (K, N), device=dev).bfloat16()
            C = torch.empty(M, N, device=dev, dtype=cdt)
            for _ in range(20):
                torch.matmul(heat_a, heat_a)
            mine.append(timed(lambda: ops._gemm_bf16(A, K, 0, B, B.stride(0), tb, C, N, M, N, K)))
            if cdt == torch.bfloat16:
                Ct = torch.empty(M, N, device=dev, dtype=cdt)
                Bt = B.t() if tb else B
                lib.append(timed(lambda: torch.matmul(A, Bt, out=Ct)))
        rounds = -(-M // 256) * -(-N // 256) / 256.0
        pk, fx = np.polyfit(Ks[2:], mine[2:], 1)
        line = f"tB{tb} C={'bf16' if cdt == torch.bfloat16 else 'fp32'}  ours " + " ".join(f"{k}:{u:6.1f}" for k, u in zip(Ks, mine)) + \
               f" | fixed {fx:6.1f} us ({fx / rounds:5.2f}/round) per-64k {pk * 64:5.2f} us ({pk * 64 / rounds:5.3f}/round)"
        if lib:
            pk2, fx2 = np.polyfit(Ks[2:], lib[2:], 1)
            line += "\n            lib  " + " ".join(f"{k}:{u:6.1f}" for k, u in zip(Ks, lib)) + \
                    f" | fixed {fx2:6.1f} us ({fx2 / rounds:5.2f}/round) per-64k {pk2 * 64:5.2f} us ({pk2 * 64 / rounds:5.3f}/round)"
        print(line, flush=True)
