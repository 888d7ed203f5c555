"""Which kernels does the vendor library behind torch.matmul run for the cfg-2 fp32 shapes?  (run under rocprofv3 --kernel-trace --stats)"""
import torch
dev = torch.device("cuda", 0)
shapes = [(16128, 1024, 1024, 0, 1), (16128, 1024, 3072, 0, 0), (1024, 1024, 16128, 1, 0), (2048, 1024, 16128, 1, 0), (3072, 1024, 16128, 1, 0),
          (4480, 3072, 768, 0, 1), (4480, 2304, 768, 0, 1), (768, 3072, 4480, 1, 0)]
for M, N, K, ta, tb in shapes:
    A = torch.randn((K, M) if ta else (M, K), device=dev)
    B = torch.randn((N, K) if tb else (K, N), device=dev)
    C = torch.empty(M, N, device=dev)
    At, Bt = (A.t() if ta else A), (B.t() if tb else B)
    for _ in range(6):
        torch.matmul(At, Bt, out=C)
    torch.cuda.synchronize()
